// Fused forward of the GNS K-step loop (GNS/main.py:140-202) for gfx950, plus the two layout kernels
// in front of it.  One launch runs all K steps of 64*gridDim.x grids.
#include "gns_device.h"
#ifndef GNS_FWD_GEN_SKIP
#define GNS_FWD_GEN_SKIP 1          // 0 (diagnostic): L_v / phi_v are computed on generator buses too (their results are discarded)
#endif
#include "gns_kernels.h"

// ------------------------------------------------------------------------------------------------
// pack_params: flat state_dict-order parameters -> folded T-stream (forward) and N-stream (backward) blocks
// (gns_common.h: phi' = first two layers of phi; L' = L with the phi output layer folded into its first layer)
// ------------------------------------------------------------------------------------------------
__global__ void gns_pack_params_kernel(const float* __restrict__ flat, float* __restrict__ pt, float* __restrict__ pn,
                                       GnsFamilies fam, int K, int D, int H) {
  // D, H = the kernel's dims (destination layouts); Dr, Hr <= them = the model's (source layout).  Positions the model does not
  // have are written as zeros (gns_common.h, GnsFamilies: a narrower model runs zero-padded).
  const int blk = blockIdx.x;            // (family, k)
  const int f = blk / K, k = blk % K;
  const bool is_phi = f < fam.nphi;
  const int Dr = fam.dr, Hr = fam.hr;
  const int INr = fam.in[f], OUTr = fam.out[f];
  const int OUT = fam.outk[f], OUTP = OUT + (OUT & 1);
  const float* src = flat + fam.flat_off[f] + (int64_t)k * fam.flat_sz[f];
  const int sW1 = 0, sb1 = INr * Hr, sW2 = sb1 + Hr, sb2 = sW2 + Hr * Hr, sW4 = sb2 + Hr, sb4 = sW4 + OUTr * Hr;
  auto W2r = [&](int j, int i) -> float { return (j < Hr && i < Hr) ? src[sW2 + j * Hr + i] : 0.f; };      // linear2.weight[j][i]
  auto b1r = [&](int j) -> float { return j < Hr ? src[sb1 + j] : 0.f; };
  auto b2r = [&](int j) -> float { return j < Hr ? src[sb2 + j] : 0.f; };
  float* t = pt + fam.t_off[f] + (int64_t)k * fam.t_sz[f];
  float* n = pn + fam.n_off[f] + (int64_t)k * fam.n_sz[f];
  if (is_phi) {
    const int IN = D + 5, INP = IN + (IN & 1);
    // input i of the kernel's phi': latent component i (< D) or line parameter i - D; -1 = a padded latent component
    auto W1r = [&](int j, int i) -> float {                       // linear1.weight[j][i] in kernel indices
      if (j >= Hr) return 0.f;
      const int ir = i < D ? (i < Dr ? i : -1) : Dr + (i - D);
      return ir >= 0 ? src[sW1 + j * INr + ir] : 0.f;
    };
    const int ob1 = IN * H, oW2 = ob1 + H, ob2 = oW2 + H * H, tt = ob2 + H;
    for (int e = threadIdx.x; e < (int)fam.t_sz[f]; e += blockDim.x) {
      float v = 0.f;
      if (e < ob1) { int i = e / H, j = e % H; v = W1r(j, i); }
      else if (e < oW2) v = b1r(e - ob1);
      else if (e < ob2) { int q = e - oW2, i = q / H, j = q % H; v = W2r(j, i); }
      else if (e < tt) v = b2r(e - ob2);
      t[e] = v;
    }
    const int nW1 = H * H, nt = nW1 + H * INP;
    for (int e = threadIdx.x; e < (int)fam.n_sz[f]; e += blockDim.x) {
      float v = 0.f;
      if (e < nW1) v = W2r(e / H, e % H);
      else if (e < nt) { int q = e - nW1, j = q / INP, i = q % INP; v = i < IN ? W1r(j, i) : 0.f; }
      else { int q = e - nt, g = q / (4 * H), j = (q % (4 * H)) / 4, i = 4 * g + (q & 3); v = (g < (D + 3) / 4 && i < D) ? W1r(j, i) : 0.f; }   // W1x: latent columns
      n[e] = v;
    }
    return;
  }
  // L family: input of the folded first layer = [v theta dp dq | m (D) | sum_e h_e (H) | deg]
  const int fp = fam.phi_of[f];
  const int PO = fam.out[fp];                                   // phi output width of the model: Dr (multi) or 1
  const float* ps = flat + fam.flat_off[fp] + (int64_t)k * fam.flat_sz[fp];
  const int PINr = fam.in[fp];
  const float* pW4 = ps + PINr * Hr + Hr + Hr * Hr + Hr;        // phi linear4.weight [PO][Hr]
  const float* pb4 = pW4 + PO * Hr;                             // phi linear4.bias  [PO]
  const int HEAD = 4 + D, HEADr = 4 + Dr, INF = HEAD + H + 1, INFP = INF + (INF & 1);
  auto w1f = [&](int c, int i) -> float {                       // folded first-layer weight W1'[c][i] in kernel indices
    if (c >= Hr) return 0.f;
    if (i < HEAD) return (i < 4 || i - 4 < Dr) ? src[sW1 + c * INr + i] : 0.f;     // (4 + latent index: the same offset in both layouts)
    float acc = 0.f;
    if (i < HEAD + H) { const int j = i - HEAD; if (j < Hr) for (int q = 0; q < PO; ++q) acc += src[sW1 + c * INr + HEADr + q] * pW4[q * Hr + j]; }
    else if (i == HEAD + H) { for (int q = 0; q < PO; ++q) acc += src[sW1 + c * INr + HEADr + q] * pb4[q]; }
    return acc;
  };
  auto W4r = [&](int j, int i) -> float { return (j < OUTr && i < Hr) ? src[sW4 + j * Hr + i] : 0.f; };    // linear4.weight[j][i]
  const int ob1 = INF * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW4 = ob2 + H, ob4 = oW4 + H * OUTP, tt = ob4 + OUTP;
  for (int e = threadIdx.x; e < (int)fam.t_sz[f]; e += blockDim.x) {
    float v = 0.f;
    if (e < ob1) { int i = e / H, j = e % H; v = w1f(j, i); }
    else if (e < oW2) v = b1r(e - ob1);
    else if (e < ob2) { int q = e - oW2, i = q / H, j = q % H; v = W2r(j, i); }
    else if (e < oW4) v = b2r(e - ob2);
    else if (e < ob4) { int q = e - oW4, i = q / OUTP, j = q % OUTP; v = W4r(j, i); }
    else if (e < tt) { int j = e - ob4; v = j < OUTr ? src[sb4 + j] : 0.f; }
    t[e] = v;
  }
  const int nW2 = OUTP * H, nW1 = nW2 + H * H, nt = nW1 + H * INFP;
  for (int e = threadIdx.x; e < (int)fam.n_sz[f]; e += blockDim.x) {
    float v = 0.f;
    if (e < nW2) { int j = e / H, i = e % H; v = W4r(j, i); }
    else if (e < nW1) { int q = e - nW2; v = W2r(q / H, q % H); }
    else if (e < nt) { int q = e - nW1, j = q / INFP, i = q % INFP; v = i < INF ? w1f(j, i) : 0.f; }
    else { int q = e - nt, g = q / (4 * H), j = (q % (4 * H)) / 4, i = 4 * g + (q & 3); v = (g < (INF + 3) / 4 && i < INF) ? w1f(j, i) : 0.f; }   // W1x
    n[e] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// pack_inputs: reference layout [Bt,N,6] / [Bt,E,7] / [Bt,Gn,7] -> float4 rows [group][row][lane]
// and the batch-invariant pieces of GNS.forward's prologue (main.py:144-152) and of the physics
// (y = 1/sqrt(r^2+x^2), main.py:38,87; the bus-id-as-line-index gathers of y, tau, shift, b).
// ------------------------------------------------------------------------------------------------
// One wave per unit, units of one kind per workgroup: a bus (its 3 rows), a line in dst order (its 3 rows), four
// lines in src order (1 row each), the per-grid sums.  Lane = grid, so every load of a wave touches 64 different cache
// lines and the kernel is latency-bound: a unit issues all its loads before it uses any, and the [.,6] / [.,7] tables
// (rows only 4-byte aligned) are read 2-4 columns at a time (gfx950 global loads need dword alignment only).
__global__ void gns_pack_inputs_kernel(const int* __restrict__ topo, const float* __restrict__ buses,
                                       const float* __restrict__ lines, const float* __restrict__ gens,
                                       float* __restrict__ out, int N, int E, int Gn, long long Bt, long long rows, long long groups) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2; every workgroup of one group of 64 grids
  // re-reads the same 0.6 MB of input, so a group is kept on one XCD: linear id -> (xcd, j), group = 8 (j / chunks) + xcd.
  const int cb = (N + W - 1) / W, ci = (E + W - 1) / W, co = ((E + 3) / 4 + W - 1) / W;
  const long long chunks = cb + ci + co + 1;
  const long long lin = blockIdx.x, j = lin >> 3;
  const long long g = (j / chunks) * 8 + (lin & 7);
  const int c = (int)(j % chunks);
  if (g >= groups) return;
  long long b = g * GNS_LANES + lane;
  if (b >= Bt) b = Bt - 1;               // dead lanes replay the last grid; their results are never stored
  const float* bu = buses + b * (long long)N * 6;
  const float* li = lines + b * (long long)E * 7;
  const float* ge = gens + b * (long long)Gn * 7;
  struct __attribute__((packed, aligned(4))) U4 { float x, y, z, w; };
  struct __attribute__((packed, aligned(4))) U2 { float x, y; };
  auto ld4 = [](const float* p) { const U4 u = *reinterpret_cast<const U4*>(p); return f4{u.x, u.y, u.z, u.w}; };
  auto ld2 = [](const float* p) { const U2 u = *reinterpret_cast<const U2*>(p); return f2{u.x, u.y}; };
  auto yof = [](float r, float x) {      // main.py:38: 1 / sqrt(r^2 + x^2), each op rounded like torch does
    return __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(__fmul_rn(r, r), __fmul_rn(x, x))));
  };
  float* og = out + g * rows * (GNS_LANES * 4);
  auto put = [&](long long row, const f4& o) { *row_ptr_lanewise(og, row, lane) = o; };

  if (c < cb) {                                                     // ---- bus n: rows 3n .. 3n+2
    const int n = c * W + wave;
    if (n >= N) return;
    const f4 bq = ld4(bu + n * 6 + 2);                              // Pd, Qd, Gs, Bs
    float pmin = 0.f, pset = 0.f, pmax = 0.f, vg = 0.f, pg = 0.f, qg = 0.f;
    const int g0 = topo[topo[TH_GEN_PTR] + n], g1 = topo[topo[TH_GEN_PTR] + n + 1];
    for (int q = g0; q < g1; ++q) {
      const float* r = ge + topo[topo[TH_GEN_IDX] + q] * 7;         // (bus_i,Pmax,Pmin,Pg_set,vg,qg,Pg) utils.py:9
      const f4 ra = ld4(r + 1);
      const f2 rb = ld2(r + 5);
      pmax += ra.x; pmin += ra.y; pset += ra.z; vg += ra.w; qg += rb.x; pg += rb.y;
    }
    const float v0 = (vg == 0.f) ? 1.f : vg;                        // main.py:146-147
    put(3LL * n, bq);
    put(3LL * n + 1, f4{pmin, pset, pmax, bq.z});                   // + Gs: all the backward's Pb-0 needs of a bus, in one row
    put(3LL * n + 2, f4{__fsub_rn(__fsub_rn(pg, bq.x), __fmul_rn(bq.z, __fmul_rn(v0, v0))),          // main.py:150
                        __fadd_rn(__fsub_rn(qg, bq.y), __fmul_rn(bq.w, __fmul_rn(v0, v0))), v0, 0.f});   // main.py:152
  } else if (c < cb + ci) {                                         // ---- line p in dst order: rows 3N + 3p ..
    const int p = (c - cb) * W + wave;
    if (p >= E) return;
    const int e = topo[topo[TH_IN_EID] + p], s = topo[topo[TH_IN_SRC] + p];
    const f4 ea = ld4(li + e * 7 + 2);                              // r, x, b, tau
    const float she = li[e * 7 + 6];
    const f4 sa = ld4(li + s * 7 + 2);                              // line NUMBER s: r, x, b, tau
    const float shs = li[s * 7 + 6];
    put(3LL * N + 3LL * p, ea);
    put(3LL * N + 3LL * p + 1, f4{she, yof(sa.x, sa.y), sa.w, shs});
    put(3LL * N + 3LL * p + 2, f4{sa.z, 0.f, 0.f, 0.f});
  } else if (c < cb + ci + co) {                                    // ---- four lines in src order: rows 3N + 3E + q
    const int q0 = ((c - cb - ci) * W + wave) * 4;
    f4 ta[4];
    float sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = topo[topo[TH_OUT_DST] + min(q0 + i, E - 1)];
      ta[i] = ld4(li + t * 7 + 2);
      sh[i] = li[t * 7 + 6];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (q0 + i < E) put(3LL * N + 3LL * E + q0 + i, f4{yof(ta[i].x, ta[i].y), ta[i].w, sh[i], ta[i].z});
  } else {                                                          // ---- per-grid sums (main.py:45,47-51)
    if (wave != 0) return;
    float sPd = 0.f, sset = 0.f, smin = 0.f, smax = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) sPd += bu[n * 6 + 2];
#pragma unroll 4
    for (int q = 0; q < Gn; ++q) { const f2 pm = ld2(ge + q * 7 + 1); smax += pm.x; smin += pm.y; sset += ge[q * 7 + 3]; }
    put(3LL * N + 4LL * E, f4{sPd, sset, smin, smax});
  }
}

// ------------------------------------------------------------------------------------------------
// The fused forward kernel.
#ifndef GNS_FWD_SKIP_LAST_M
#define GNS_FWD_SKIP_LAST_M 1      // 0 (diagnostic): run the dead m-family units of the last step like the reference does
#endif
#ifndef GNS_FWD_ROW_STORES
#define GNS_FWD_ROW_STORES 0      // 1 (diagnostic): keep the dead partial stores of theta / v into the state row
#endif
// ------------------------------------------------------------------------------------------------
template <int D, int H, bool MULTI>
__global__ void __launch_bounds__(GNS_FWD_MAX_THREADS) gns_forward_kernel(GnsFwdArgs A) {
  using C = GnsDims<D, H, MULTI>;
  constexpr int MQ = C::MQ, RB = C::RB;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  // team of A.team workgroups per 64-grid group (gns_device.h): blocks 8 apart share an XCD (and its L2) when the groups allow it
  const int tsize = A.team;
  long long g = blockIdx.x;
  int member = 0;
  if (tsize > 1) {
    if ((A.G & 7) == 0) { g = (long long)(blockIdx.x / (8 * tsize)) * 8 + (blockIdx.x & 7); member = (blockIdx.x >> 3) % tsize; }
    else { g = blockIdx.x / tsize; member = blockIdx.x % tsize; }
  }
  const int cw = member * nwaves + wave, tw = tsize * nwaves;       // this wave among the waves of the group
  const int N = A.N, E = A.E, K = A.K;
  cip topo = (cip)A.topo;
  cfp PT = (cfp)A.pt;
  const cip in_ptr = topo + topo[TH_IN_PTR], in_src = topo + topo[TH_IN_SRC], in_a = topo + topo[TH_IN_A],
            in_b = topo + topo[TH_IN_B], out_ptr = topo + topo[TH_OUT_PTR], out_dst = topo + topo[TH_OUT_DST],
            out_c = topo + topo[TH_OUT_C], out_d = topo + topo[TH_OUT_D], is_gen = topo + topo[TH_IS_GEN],
            part = topo + topo[TH_PART] + A.part_idx * (GNS_MAXP + 1);
  const int n0 = part[cw], n1 = part[cw + 1];
  const cip upart = topo + topo[TH_UPART] + A.part_idx * (GNS_MAXP + 1), ppart = topo + topo[TH_PPART] + A.part_idx * (GNS_MAXP + 1);
  const int u0 = upart[cw], u1 = upart[cw + 1];           // (family group, bus) units of the update phase, group-major
  const int q0w = ppart[cw], q1w = ppart[cw + 1];         // buses of the physics phase
  const long long R = gns_in_rows(N, E);
  const float* IN = A.in;
  const long long in_base = g * R;
  const long long row_ein = in_base + 3LL * N, row_eout = row_ein + 3LL * E, row_grid = row_eout + E;
  const long long b = g * GNS_LANES + lane;
  const bool live = b < A.Bt;

  // (v, theta) of every bus of the 64 grids for the step being produced, written by the update phase and gathered by
  // the line physics (6 neighbour buses per line): 60 KB for case118 instead of ~19 HBM rows per bus and step.
  extern __shared__ __attribute__((aligned(16))) unsigned char gns_dyn_lds[];
  f2* plane = reinterpret_cast<f2*>(gns_dyn_lds);
  const bool use_plane = A.plane != 0;
  // A.plane == 2 (one workgroup per group and room for it): a second plane holds (delta_p before the generator term, delta_q)
  // between the physics phase and the lambda phase of a step - the same wave owns a bus in both - so the state row is written
  // once per step instead of written, re-read and written again
  const bool use_plane2 = A.plane == 2;
  f2* plane2 = plane + (size_t)N * GNS_LANES;
  // per-wave partial sums [2 parities][GNS_MAXW][64][2] behind the planes; a team keeps them in HBM and the LDS goes to the plane
  float* red = reinterpret_cast<float*>(gns_dyn_lds + (use_plane ? (size_t)N * GNS_LANES * sizeof(f2) : 0) * (use_plane2 ? 2 : 1));
  __shared__ int unit_ctr[2];                        // evaluation mode: work queue of the update phase, one counter per step parity
  __shared__ int team_failed;
  if (threadIdx.x < 2) unit_ctr[threadIdx.x] = 0;
  if (threadIdx.x == 0) team_failed = 0;
  GnsTeam team;
  team.size = tsize; team.member = member; team.epoch = 0; team.failed = &team_failed;
  team.ctr = reinterpret_cast<unsigned*>(A.team_ws + g * GNS_TEAM_CTR_BYTES);
  team.red = reinterpret_cast<float*>(A.team_ws + A.G * GNS_TEAM_CTR_BYTES) + g * GNS_TEAM_RED_FLOATS;
  __syncthreads();                                   // team_failed is initialised
  team_setup(team, team.ctr);
  // per-wave partial sums: LDS inside one workgroup, the group's HBM buffer across a team
  auto red_put = [&](int par, float a, float c) {
    if (tsize == 1) *reinterpret_cast<f2*>(red + ((par * GNS_MAXW + wave) * GNS_LANES + lane) * 2) = f2{a, c};
    else reinterpret_cast<f2*>(team.red)[(par * GNS_MAXP + cw) * GNS_LANES + lane] = f2{a, c};
  };
  auto red_sum = [&](int par, float& a, float& c) {
    a = 0.f; c = 0.f;
    if (tsize == 1) { for (int w = 0; w < nwaves; ++w) { const f2 r = *reinterpret_cast<const f2*>(red + ((par * GNS_MAXW + w) * GNS_LANES + lane) * 2); a += r.x; c += r.y; } }
    else for (int w = 0; w < tw; ++w) { const f2 r = reinterpret_cast<const f2*>(team.red)[(par * GNS_MAXP + w) * GNS_LANES + lane]; a += r.x; c += r.y; }
  };

  auto state_row = [&](int slot, int n) { return (((long long)slot * A.G + g) * N + n) * RB; };

  // ---- prologue (main.py:141-152): m = 0, theta = 0, v = vg or 1, delta_p/q from the set points
  for (int n = n0; n < n1; ++n) {
    const f4 b2 = *row_ptr(IN, in_base + 3LL * n + 2, lane);       // dp0, dq0, v0
    const long long r0 = state_row(0, n);
    *row_ptr(A.state, r0, lane) = f4{b2.z, 0.f, b2.x, b2.y};        // v0, 0, dp0, dq0
#pragma unroll
    for (int q = 0; q < MQ; ++q) *row_ptr(A.state, r0 + 1 + q, lane) = f4{0.f, 0.f, 0.f, 0.f};
  }
  team_barrier(team);                               // the update phase reads buses initialised by other waves
  const f4 gsum = *row_ptr(IN, row_grid, lane);     // (sumPd, sumPset, sumPmin, sumPmax)
  float tot_part = 0.f, last_part = 0.f;
  const float invN = 1.0f / (float)N;

#ifdef GNS_STAMPS
  long long tph[6] = {0, 0, 0, 0, 0, 0};
  long long tlast = clock64();
#define FSTAMP(i) { const long long tn = clock64(); tph[i] += tn - tlast; tlast = tn; }
#else
#define FSTAMP(i)
#endif
  for (int k = 0; k < K; ++k) {
    const int rs = A.save ? k : (k & 1), ws = A.save ? k + 1 : ((k + 1) & 1);
    FSTAMP(5)
    const long long koff = (long long)k;
    // (warming the scalar cache with this step's weights here - gns_device.h, scalar_cache_warm - changes nothing: 0.841-0.846 ms
    //  with, 0.843-0.845 ms without; sixteen waves hide the cold lines of a step's first units)
    // ================= phase U: latent / v / theta update (main.py:155-188) =================================
    // Work unit = (family, bus): family theta writes theta, family v writes v, family m writes the latent vector.
    // Units are dealt family-major so that a wave streams one family's weights (scalar-cache resident) and the 16
    // waves balance to a few per cent (whole buses of 750..5500 instructions left 16 % of the time at the barrier).
    // unit group 0 = the theta and v families of one bus together (one load of the bus state and of the line rows,
    // 8.9 KB of weights), unit group 1 = the m family.  The forward is HBM-bound in training mode, so re-reading the
    // state once per family costs more than the second family's weights in the scalar cache.
    auto update_unit = [&](auto grp_, int n) {
      constexpr int grp = decltype(grp_)::value;
      constexpr int NL = grp == 0 ? 2 : 1, L0 = grp == 0 ? 0 : 2;      // L_theta, L_v | L_m   (main.py:173-180)
      const long long rr = state_row(rs, n), wr = state_row(ws, n);
      const f4 s0 = *row_ptr(A.state, rr, lane);
      f2 m[D / 2];
      if (k == 0) {                                         // m_0 = 0 (main.py:141): known, not read back
#pragma unroll
        for (int i = 0; i < D / 2; ++i) m[i] = f2{0.f, 0.f};
      } else {
        load_pairs<D>(A.state, rr + 1, lane, m);
      }
      const int p0 = in_ptr[n], p1 = in_ptr[n + 1];
      f2 S[NL][H / 2];                                      // sum over the lines ending at n of the hidden vector of phi'
#pragma unroll
      for (int j = 0; j < NL; ++j)
#pragma unroll
        for (int q = 0; q < H / 2; ++q) S[j][q] = f2{0.f, 0.f};
      // main.py:155-163 with the output layer of phi folded into L'.  The latent vector of the destination bus is the
      // same for every line ending here: its share of phi's first layer is computed once per bus (phi_head)
      constexpr int NF = MULTI ? NL : 1;
#if GNS_FWD_GEN_SKIP
      // v moves only on buses without a generator (main.py:184-186): on a generator bus L_v's output is discarded, and with three phi
      // nets so is everything phi_v computes for the lines ending there; the branch is uniform (one topology per wave).  The hidden sum
      // is then saved as zeros (the persistent backward kernels still multiply it by a zero upstream; the split backward skips it too).
      const bool skip_v = grp == 0 && is_gen[n] != 0;
#else
      const bool skip_v = false;
#endif
      f2 uh[NF][H / 2];
      if (p0 < p1) {
        static_for<0, NF>([&](auto j_) {
          constexpr int j = decltype(j_)::value;
          constexpr int l = L0 + j;
          constexpr int fphi = MULTI ? (l == 0 ? 1 : (l == 1 ? 0 : 2)) : 0;   // phi_theta, phi_v, phi_m
          if (MULTI && l == 1 && skip_v) return;
          phi_head<D, H>(PT + A.t_off[fphi] + koff * A.t_sz[fphi], m, uh[j]);
        });
      }
      for (int p = p0; p < p1; ++p) {
        const f4 e0 = *row_ptr(IN, row_ein + 3LL * p, lane), e1 = *row_ptr(IN, row_ein + 3LL * p + 1, lane);
        const f2 xt[3] = {f2{e0.x, e0.y}, f2{e0.z, e0.w}, f2{e1.x, 0.f}};          // r, x, b, tau, shift
        static_for<0, NF>([&](auto j_) {
          constexpr int j = decltype(j_)::value;
          constexpr int l = L0 + j;
          constexpr int fphi = MULTI ? (l == 0 ? 1 : (l == 1 ? 0 : 2)) : 0;
          if (MULTI && l == 1 && skip_v) return;
          f2 a1[H / 2], a2[H / 2];
          phi_tail<C::PHI_IN, H, D>(PT + A.t_off[fphi] + koff * A.t_sz[fphi], uh[j], xt, a1, a2);
#pragma unroll
          for (int q = 0; q < H / 2; ++q) S[j][q] += a2[q];
        });
      }
      f2 vth_new = f2{s0.x, s0.y};
      static_for<0, NL>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        constexpr int l = L0 + j;
        constexpr int fphi = MULTI ? (l == 0 ? 1 : (l == 1 ? 0 : 2)) : 0;
        constexpr int js = MULTI ? j : 0;                   // the single phi: one sum serves all three L nets (main.py:169-171)
#ifndef GNS_ABLATE_MSG_SAVE      // diagnostic: the hidden sums are not saved (the backward would have to recompute them): what their 1.1 GB of writes cost the forward
        if (A.save && (MULTI || l == 0))
#else
        if (false)
#endif
          store_pairs_nt<H>(A.msg, ((((long long)k * A.G + g) * N + n) * C::NPHI + fphi) * C::HQ, lane, S[js]);
        if (l == 1 && skip_v) {                             // (a generator bus keeps its voltage: main.py:184-186)
          vth_new.x = s0.x;
          if (!(use_plane && tsize == 1) || GNS_FWD_ROW_STORES) reinterpret_cast<float*>(row_ptr(A.state, wr, lane))[0] = s0.x;
          return;
        }
        f2 x[(C::LF_IN + 1) / 2];                           // [v theta | dp dq | m | sum h | deg]
        x[0] = f2{s0.x, s0.y}; x[1] = f2{s0.z, s0.w};
#pragma unroll
        for (int i = 0; i < D / 2; ++i) x[2 + i] = m[i];
#pragma unroll
        for (int q = 0; q < H / 2; ++q) x[2 + D / 2 + q] = S[js][q];
        x[2 + D / 2 + H / 2] = f2{(float)(p1 - p0), 0.f};
        f2 a1[H / 2], a2[H / 2];
        if constexpr (l < 2) {
          f2 y[1];
          mlp_fwd<C::LF_IN, H, 2>(PT + A.t_off[C::NPHI + l] + koff * A.t_sz[C::NPHI + l], x, a1, a2, y);
          // With the LDS plane inside one workgroup the physics phase takes (v, theta) from the plane and then writes the whole
          // state row: the two 4-byte-per-lane stores into that row (every 64-byte sector of 1 KiB touched for 256 bytes) are dead
          float* r0 = reinterpret_cast<float*>(row_ptr(A.state, wr, lane));
          const bool row_needed = !(use_plane && tsize == 1) || GNS_FWD_ROW_STORES;
          if constexpr (l == 0) { vth_new.y = s0.y + y[0].x; if (row_needed) r0[1] = vth_new.y; }                    // theta += L_theta        (main.py:182)
          else { vth_new.x = is_gen[n] ? s0.x : s0.x + y[0].x; if (row_needed) r0[0] = vth_new.x; }                  // v += L_v off generators (main.py:184-186)
        } else {
          f2 upd_m[D / 2], m_new[D / 2];
          mlp_fwd<C::LF_IN, H, D>(PT + A.t_off[C::NPHI + 2] + koff * A.t_sz[C::NPHI + 2], x, a1, a2, upd_m);
#pragma unroll
          for (int i = 0; i < D / 2; ++i) m_new[i] = m[i] + upd_m[i];        // main.py:188
          if (k < K - 1) store_pairs_nt<D>(A.state, wr + 1, lane, m_new);   // nothing reads m_K (the reference's L_m.{K-1} has no gradient)
        }
      });
      if constexpr (grp == 0) { if (use_plane) plane[n * GNS_LANES + lane] = vth_new; }
    };
    // Evaluation mode: the waves DRAW units from an LDS counter instead of owning a fixed range.  The SIMD arbiter
    // favours its oldest wave, so with fixed shares the youngest four waves finish the phase 2x later and run its tail
    // alone (24 % of the kernel at the barrier); drawn units end the phase together (0.915 -> 0.894 ms).  A
    // unit writes only its own bus and nothing is summed across units here, so which wave runs a unit cannot change
    // a bit of the result.  Training mode keeps the fixed ranges: with the state / hidden-sum saves in flight the
    // drawn order measured 4 % slower.
    const bool draw = A.save == 0 && tsize == 1;
    // The last step's m family is dead work: m_K feeds nothing (the reference's L_m.{K-1} / phi_m.{K-1} get no gradient for
    // the same reason) and its hidden sums are only read by the backward sweep that is skipped.  The last step therefore
    // runs the (theta, v) units only, dealt over ALL waves by the bus ranges (units [n0, n1) are group 0): 1/K of the m
    // family's instructions and rows, and the waves that owned m units help with the rest.
    const bool last_step = k == K - 1 && GNS_FWD_SKIP_LAST_M;
    const int ub = last_step ? n0 : u0, ue = last_step ? n1 : u1, uall = last_step ? N : 2 * N;
    for (int u = ub;;) {
      if (draw) {
        int t = 0;
        if (lane == 0) t = atomicAdd(&unit_ctr[k & 1], 1);
        u = __builtin_amdgcn_readfirstlane(t);
        if (u >= uall) break;
      } else if (u >= ue) break;
      const int grp = u / N, n = u - grp * N;
      if (grp == 0) update_unit(std::integral_constant<int, 0>{}, n);
      else update_unit(std::integral_constant<int, 1>{}, n);
      ++u;
    }
    FSTAMP(0)
    team_barrier(team);   // every bus of the 64 grids now has v_{k+1}, theta_{k+1}
    if (use_plane && tsize > 1) {                          // a team member produced only its share of the plane: fetch all of it (N KB from L2)
      for (int n = wave; n < N; n += nwaves) {
        const f4 r = *row_ptr(A.state, state_row(ws, n), lane);
        plane[n * GNS_LANES + lane] = f2{r.x, r.y};
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) unit_ctr[(k + 1) & 1] = 0;      // idle since the previous step; next drawn from two barriers from here
    FSTAMP(1)

    // ================= phase P: line physics (main.py:34-104), bus-centric, no scatter ================
    auto vth = [&](int n) {                                           // (v, theta)_{k+1} of any bus of this wave's 64 grids
      if (use_plane) return plane[n * GNS_LANES + lane];
      const f4 r = *row_ptr(A.state, state_row(ws, n), lane);
      return f2{r.x, r.y};
    };
    float joule = 0.f, v2gs = 0.f;
    for (int n = q0w; n < q1w; ++n) {
      const long long wr = state_row(ws, n);
      const f2 sn = vth(n);
      const float vn = sn.x, thn = sn.y;
      const f4 b0 = *row_ptr(IN, in_base + 3LL * n, lane);            // Pd,Qd,Gs,Bs
      float sum_pf = 0.f, sum_qf = 0.f, sum_pt = 0.f, sum_qt = 0.f, agg = 0.f;
      const int p0 = in_ptr[n], p1 = in_ptr[n + 1];
      for (int p = p0; p < p1; ++p) {                                 // lines with dst == n ("from" messages)
        const int s = in_src[p], ia = in_a[p], ib = in_b[p];
        const f4 e1 = *row_ptr(IN, row_ein + 3LL * p + 1, lane), e2 = *row_ptr(IN, row_ein + 3LL * p + 2, lane);
        const f2 ss = vth(s);
        const float tha = vth(ia).y, thb = vth(ib).y;
        const float vs = ss.x, ths = ss.y, vt = vn, tht = thn;
        const float ys = e1.y, taus = e1.z, shs = e1.w, bs = e2.x;
        const float dl = tha - thb;                                   // delta_ij[src]  (bus id used as line index)
        const float angA = ths - tht - dl - shs, angB = tht - ths - dl + shs;
        float sA, cA, sD, cD;
        sincosf(angA, &sA, &cA);
        sincosf(dl, &sD, &cD);
        const float sB = sinf(angB);
        const float base = vs * vt * ys / taus;
        const float msg = fabsf(base * (sA + sB) + (vs / (taus * taus)) * ys * sD + (vt * vt) * ys * sD);   // main.py:41
        agg += msg;
        const float vst = vs / taus, vst2 = vst * vst;
        sum_pf += base * sA + vst2 * ys * sD;                          // main.py:91
        sum_qf += -base * cA + vst2 * (ys * cD - bs / 2.f);            // main.py:68-69 == :98
      }
      const int q0 = out_ptr[n], q1 = out_ptr[n + 1];
      for (int q = q0; q < q1; ++q) {                                 // lines with src == n ("to" messages)
        const int t = out_dst[q], ic = out_c[q], id = out_d[q];
        const f4 o0 = *row_ptr(IN, row_eout + q, lane);               // y_t, tau_t, sh_t, b_t
        const f2 st = vth(t);
        const float thc = vth(ic).y, thd = vth(id).y;
        const float vs = vn, ths = thn, vt = st.x, tht = st.y;
        const float dl = thd - thc;                                   // delta_ji[dst]
        const float angC = tht - ths - dl - o0.z;
        float sC, cC;
        sincosf(angC, &sC, &cC);
        const float sD = sinf(dl);
        const float base = vt * vs * o0.x / o0.y;
        sum_pt += base * sC + (vt * vt) * o0.x * sD;                   // main.py:92
        sum_qt += -base * cC + (vt * vt) * (o0.x * sD - o0.w / 2.f);   // main.py:70-72 == :99
      }
      joule += agg;
      const float v2 = vn * vn;
      v2gs += v2 * b0.z;
      const float dp_pre = ((0.f - b0.x) - b0.z * v2) + sum_pf + sum_pt;     // main.py:82,96 without the generator term
      const float qg_new = ((b0.y - b0.w * v2) - sum_qf) - sum_qt;           // main.py:64,76
      const float dq = ((qg_new - b0.y) + b0.w * v2) + sum_qf + sum_qt;      // main.py:83,103 (cancels to rounding noise)
      if (use_plane2) plane2[n * GNS_LANES + lane] = f2{dp_pre, dq};
      else *row_ptr(A.state, wr, lane) = f4{vn, thn, dp_pre, dq};
    }
    FSTAMP(2)
    red_put(k & 1, joule, v2gs);
    team_barrier(team);
    FSTAMP(3)
    float jsum, vsum;
    red_sum(k & 1, jsum, vsum);
    // global active compensation (main.py:45-57)
    const float p_global = (gsum.x + vsum) + jsum;
    float lam;
    const bool low1 = p_global < gsum.y;
    if (low1) lam = (p_global - gsum.z) / (2.f * (gsum.y - gsum.z));
    else lam = (p_global - 2.f * gsum.y + gsum.w) / (2.f * (gsum.w - gsum.y));
    const bool low2 = lam < 0.5f;
    if (A.save && cw == 0)
      reinterpret_cast<f2*>(A.lam)[((long long)k * A.G + g) * GNS_LANES + lane] = f2{lam, (low1 ? 1.f : 0.f) + (low2 ? 2.f : 0.f)};
    float sq = 0.f;
    for (int nb = q0w; nb < q1w; nb += 4) {                           // four buses per round: their 8 row loads fly together
      f4 b1[4], sn[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = min(nb + j, q1w - 1);
        b1[j] = *row_ptr(IN, in_base + 3LL * n + 1, lane);            // Pmin,Pset,Pmax summed per bus (, Gs)
        if (use_plane2) {
          const f2 vt_ = plane[n * GNS_LANES + lane], pq_ = plane2[n * GNS_LANES + lane];
          sn[j] = f4{vt_.x, vt_.y, pq_.x, pq_.y};
        } else {
          sn[j] = *row_ptr(A.state, state_row(ws, n), lane);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (nb + j < q1w) {
          const float pg = low2 ? b1[j].x + 2.f * (b1[j].y - b1[j].x) * lam
                                : 2.f * b1[j].y - b1[j].z + 2.f * (b1[j].z - b1[j].y) * lam;      // main.py:53-57
          sn[j].z = pg + sn[j].z;                                     // main.py:81-82,96
          *row_ptr(A.state, state_row(ws, nb + j), lane) = sn[j];
          sq += sn[j].z * sn[j].z + sn[j].w * sn[j].w;
        }
      }
    }
    tot_part += A.gw[k] * (sq * invN);           // main.py:198
    last_part = sq * invN;                                            // main.py:199
    team_barrier(team);                                               // dp_{k+1} is complete before any wave starts step k+1
    FSTAMP(4)
  }
#ifdef GNS_STAMPS
  if (lane == 0) for (int i = 0; i < 6; ++i) A.lam[((long long)blockIdx.x * nwaves + wave) * 6 + i] = (float)tph[i];   // diagnostic build: the lambda buffer is dead by now (eval mode)
#endif

  // ---- epilogue: outputs (main.py:199-202) ---------------------------------------------------------
  const int fs = A.save ? K : (K & 1);
  if (live) {
    for (int n = n0; n < n1; ++n) {
      const f4 sn = *row_ptr(A.state, state_row(fs, n), lane);
      A.v_out[b * N + n] = (sn.x < 0.f) ? 0.f : sn.x;                 // main.py:201
      A.theta_out[b * N + n] = sn.y;
    }
  }
  team_barrier(team);
  red_put(0, tot_part, last_part);
  team_barrier(team);
  if (cw == 0 && live) {
    float t, l;
    red_sum(0, t, l);
    if (team_failed) t = l = __builtin_nanf("");                     // a team barrier gave up: fail loudly
    A.total_out[b] = t;
    A.last_out[b] = l;
  }
  // ... and visibly: the status word of the team workspace (word 3 of its first counter line, zeroed with the counters before
  // the launch) is what gns_team_status hands to the host, so that no gradient of these losses is ever applied
  if (tsize > 1 && team_failed && threadIdx.x == 0)
    __hip_atomic_store(reinterpret_cast<unsigned*>(A.team_ws) + GNS_TEAM_STATUS_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static size_t fwd_dyn_lds(const GnsFwdArgs& A) {
  return (A.plane ? (size_t)A.N * GNS_LANES * sizeof(f2) : 0) * (A.plane == 2 ? 2 : 1) + (A.team == 1 ? (size_t)GNS_FWD_RED_BYTES : 0);
}

template <int D, int H, bool MULTI>
static int launch_forward_t(const GnsFwdArgs& A, int threads, hipStream_t st) {
  // (> 64 KB of dynamic LDS needs an opt-in per kernel and device: gns_fwd_init_device, once, from the library's initialisation)
  hipLaunchKernelGGL((gns_forward_kernel<D, H, MULTI>), dim3((unsigned)(A.G * A.team)), dim3(threads), fwd_dyn_lds(A), st, A);
  return hipGetLastError() == hipSuccess ? GNS_OK : GNS_ELAUNCH;
}

// One-time device initialisation (called under the library's thread-safe one-time initialisation, never inside a capture)
int gns_fwd_init_device() {
  int rc = GNS_OK;
#define GNS_CASE(DD, HH)                                                                                                              \
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gns_forward_kernel<DD, HH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, GNS_FWD_DYN_LDS_MAX) != hipSuccess) rc = GNS_ELAUNCH;   \
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gns_forward_kernel<DD, HH, false>), hipFuncAttributeMaxDynamicSharedMemorySize, GNS_FWD_DYN_LDS_MAX) != hipSuccess) rc = GNS_ELAUNCH;
  GNS_FOR_EACH_DIMS(GNS_CASE)
#undef GNS_CASE
  if (rc != GNS_OK) (void)hipGetLastError();
  return rc;
}

// Workgroups of this launch configuration that one CU holds at once (teams need every member resident: gns_api.hip)
int gns_fwd_blocks_per_cu(int d, int h, int multi, const GnsFwdArgs& A, int threads) {
  int nb = 0;
#define GNS_CASE(DD, HH)                                                                                                              \
  if (d == DD && h == HH) {                                                                                                           \
    const hipError_t e = multi ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gns_forward_kernel<DD, HH, true>, threads, fwd_dyn_lds(A))     \
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gns_forward_kernel<DD, HH, false>, threads, fwd_dyn_lds(A));   \
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }                                                                       \
    return nb;                                                                                                                        \
  }
  GNS_FOR_EACH_DIMS(GNS_CASE)
#undef GNS_CASE
  return 0;
}

int gns_launch_forward(int d, int h, int multi, const GnsFwdArgs& A, int threads, hipStream_t st) {
#define GNS_CASE(DD, HH)                                                                     \
  if (d == DD && h == HH) return multi ? launch_forward_t<DD, HH, true>(A, threads, st)      \
                                       : launch_forward_t<DD, HH, false>(A, threads, st);
  GNS_FOR_EACH_DIMS(GNS_CASE)
#undef GNS_CASE
  return GNS_EUNSUPPORTED;
}

int gns_launch_pack_params(const float* flat, float* pt, float* pn, const GnsFamilies& fam, int K, int D, int H, hipStream_t st) {
  hipLaunchKernelGGL(gns_pack_params_kernel, dim3(fam.nfam * K), dim3(256), 0, st, flat, pt, pn, fam, K, D, H);
  return hipGetLastError() == hipSuccess ? GNS_OK : GNS_ELAUNCH;
}

int gns_launch_pack_inputs(const int* topo, const float* buses, const float* lines, const float* gens, float* out, int N,
                           int E, int Gn, long long Bt, long long groups, hipStream_t st) {
  const long long rows = gns_in_rows(N, E);
  const int W = 4;
  const long long chunks = (N + W - 1) / W + (E + W - 1) / W + ((E + 3) / 4 + W - 1) / W + 1, blocks = chunks * ((groups + 7) / 8 * 8);
  if (blocks > 0x7fffffffLL) return GNS_ESIZE;
  hipLaunchKernelGGL(gns_pack_inputs_kernel, dim3((unsigned)blocks), dim3(64 * W), 0, st, topo, buses, lines, gens, out, N, E, Gn, Bt, rows, groups);
  return hipGetLastError() == hipSuccess ? GNS_OK : GNS_ELAUNCH;
}
