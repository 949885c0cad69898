// Shared host/device definitions for the MI355X GNS hot path (gfx950 only).
//
// Execution model: one LANE owns one grid, one 64-lane wave owns 64 grids, and the
// waves of a workgroup split the BUSES of those 64 grids.  Topology is identical for every grid of a
// batch, so every bus/line index is wave-uniform: indices and MLP weights travel through the scalar
// unit (s_load -> SGPR operands of v_pk_fma_f32), per-grid data is laid out [row][64 lanes][4 floats]
// so that every vector access is one fully coalesced 1 KiB transaction.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GNS_HD __host__ __device__
#else
#define GNS_HD
#endif

#define GNS_LANES 64
#define GNS_TOPO_MAGIC 0x474e5332  // "GNS2"
#define GNS_NPART 8                // bus partitions are stored for 1,2,4,8,16,32 and 12,24 parts per 64-grid group
#define GNS_MAXW 16                // waves of ONE workgroup (sizes the LDS reduction buffers)
#define GNS_MAXP 32                // waves of one 64-grid group: up to GNS_MAX_TEAM workgroups share a group (gns_device.h, "teams")
#define GNS_MAX_TEAM 4

// ---- topology blob: int32 words; hdr[i] below are word offsets from the blob start ----------------
enum {
  TH_MAGIC = 0, TH_N, TH_E, TH_GN,
  TH_IN_PTR,    // [N+1] CSR over destination bus (stable in line order)            main.py:153
  TH_IN_EID,    // [E]   original line index of in-edge p
  TH_IN_SRC,    // [E]   s = src[e]
  TH_IN_A,      // [E]   a = src[line s]   } the reference gathers delta_ij[src] : line NUMBER s=src[e]
  TH_IN_B,      // [E]   b = dst[line s]   } (main.py:41,68,91,98)
  TH_OUT_PTR,   // [N+1] CSR over source bus
  TH_OUT_EID,   // [E]
  TH_OUT_DST,   // [E]   t = dst[e]
  TH_OUT_C,     // [E]   c = src[line t]   } delta_ji[dst] : line NUMBER t=dst[e] (main.py:70-72,92,99)
  TH_OUT_D,     // [E]   d = dst[line t]
  TH_IS_GEN,    // [N]   1 if a generator sits on the bus (main.py:184-185)
  TH_GEN_PTR,   // [N+1] generators per bus, in generator order
  TH_GEN_IDX,   // [Gn]
  TH_PART,      // [GNS_NPART][GNS_MAXP+1] bus ranges per wave, balanced by work
  TH_P2Q,       // [E]   position in the source-sorted list of in-edge p
  TH_Q2P,       // [E]
  TH_EPART,     // [GNS_NPART][GNS_MAXP+1] ranges of in-edge positions per wave (backward, edge-centric)
  TH_INCD_PTR,  // [N+1] incidence list of the delta adjoints (backward)
  TH_INCD,      // [4E]  p*4 + code ; code 0:+dbar 1:-dbar 2:+dbar' 3:-dbar'
  TH_IN_DST,    // [E]   t = dst[e] of in-edge p
  TH_UPART,     // [GNS_NPART][GNS_MAXP+1] forward update phase: ranges of units u = grp*N + n (grp 0: theta+v, 1: m), balanced by work
  TH_PPART,     // [GNS_NPART][GNS_MAXP+1] forward physics phase: bus ranges balanced by incident lines
  TH_LANE_BUS,  // [N]   grid-per-workgroup mapping: bus handled by bus lane i (buses in descending in-degree order, so the
                //       lanes of one wave run similar trip counts in their incidence loops)
  TH_EREC,      // [E][8]  backward line phase: (s, t, a, b, q, c, d, 0) of in-edge p in one 32-byte record (one scalar load instead of seven)
  TH_TOTAL,     // blob length in words
  TH_HDR_WORDS = 32
};

// ---- packed per-grid inputs: float4 rows, [group][row][lane] ---------------------------------------
// bus n    : rows 3n..3n+2      (Pd,Qd,Gs,Bs) (Pmin,Pset,Pmax,Gs) (dp0,dq0,v0,0)
// in-edge p: rows 3N+3p..+2     (r,x,b,tau) (shift, y_s,tau_s,sh_s) (b_s,0,0,0)      s = src[e] used as LINE number
// out-edge q: row 3N+3E+q       (y_t,tau_t,sh_t,b_t)                                 t = dst[e] used as LINE number
// grid     : row 3N+4E          (sumPd, sumPset, sumPmin, sumPmax)
GNS_HD static inline int64_t gns_in_rows(int N, int E) { return 3LL * N + 4LL * E + 1; }

static inline int gns_part_index(int waves) {
  switch (waves) { case 1: return 0; case 2: return 1; case 4: return 2; case 8: return 3; case 16: return 4; case 32: return 5;
                   case 12: return 6; case 24: return 7; default: return -1; }   // 12, 24: bus chunks of the split backward (three one-wave workgroups per SIMD)
}

// "Teams": when a batch has fewer 64-grid groups than the chip has CUs, up to GNS_MAX_TEAM workgroups (on different CUs) share
// one group - their waves split its buses exactly as the waves of one workgroup do, meet at team barriers (a counter in HBM,
// agent-scope fences) and add their partial sums through a small HBM buffer.  `want` 0 = as many as fill `ncu` CUs.
static inline int gns_team_size(int64_t groups, int ncu, int want) {
  int c = 1;
  while (c < GNS_MAX_TEAM && groups * (c * 2) <= ncu) c *= 2;      // every workgroup of every team must be resident at once
  if (want > 0 && want < c) { c = 1; while (c * 2 <= want) c *= 2; }
  return c;
}
// per group: one 64-byte line holding the arrival counter + the partial sums [2 parities][GNS_MAXP waves][64 lanes][2]
#define GNS_TEAM_STATUS_WORD 3      // word of the FIRST group's counter line: set to 1 by a workgroup whose team barrier gave up
#define GNS_TEAM_CTR_BYTES 64
#define GNS_TEAM_RED_FLOATS (2 * GNS_MAXP * GNS_LANES * 2)
static inline size_t gns_team_bytes(int64_t groups) { return (size_t)groups * (GNS_TEAM_CTR_BYTES + (size_t)GNS_TEAM_RED_FLOATS * 4); }

// Parameter block geometry ------------------------------------------------------------------------
// flat (state_dict) block of one LearningBlock: W1[h][in] b1[h] W2[h][h] b2[h] W4[out][h] b4[out]
GNS_HD static inline int64_t gns_flat_block(int in, int h, int out) { return (int64_t)in * h + h + (int64_t)h * h + h + (int64_t)out * h + out; }
GNS_HD static inline int64_t gns_pad16(int64_t t) { return (t + 15) / 16 * 16; }

// The kernels run a FOLDED form of the networks.  The last layer of phi is linear and its output is only summed
// over the lines ending at a bus and fed to the first (linear) layer of L (main.py:155-171), so
//   W1L[:, 4+d:] . sum_e (W4 h_e + b4)  =  C . (sum_e h_e) + deg . c0,   C = W1L[:, 4+d:] W4,  c0 = W1L[:, 4+d:] b4.
// phi' = the first two layers of phi (output: the hidden vector h, h values); L' = L with the input
// [v theta dp dq | m | sum_e h_e | deg] (4 + d + h + 1 values).  28 % fewer MACs, same function up to rounding.
// T-stream (forward, weights transposed [in][out]):   phi': W1t[in][h] b1[h] W2t[h][h] b2[h]
//                                                     L'  : W1t'[in'][h] b1[h] W2t[h][h] b2[h] W4t[h][outp] b4[outp]
// N-stream (backward data path, [out][in] padded):    phi': W2n[h][h] W1n[h][inp]
//                                                     L'  : W4n[outp][h] W2n[h][h] W1n'[h][in'p]
GNS_HD static inline int gns_lin(int d, int h) { return 4 + d + h + 1; }
GNS_HD static inline int64_t gns_t_block(bool is_phi, int d, int h, int out) {
  if (is_phi) { int in = d + 5; return gns_pad16((int64_t)in * h + h + (int64_t)h * h + h); }
  int in = gns_lin(d, h), op = out + (out & 1);
  return gns_pad16((int64_t)in * h + h + (int64_t)h * h + h + (int64_t)h * op + op);
}
// The N-stream blocks end with a second copy of the first-layer weights in "input-major" order, W1x[groups of 4 inputs][H][4]
// (zero beyond the used inputs): the grid-per-workgroup backward produces the input adjoints 4 at a time from it and
// hands each finished pair on at once, instead of holding all of them until the last weight row has been streamed.
// phi' keeps only the latent columns there (the line parameters get no adjoint).
GNS_HD static inline int gns_w1x_groups(bool is_phi, int d, int h) { return is_phi ? (d + 3) / 4 : (gns_lin(d, h) + 3) / 4; }
GNS_HD static inline int64_t gns_n_w1x_off(bool is_phi, int d, int h, int out) {    // offset of W1x inside an N-stream block
  if (is_phi) { int in = d + 5, ip = in + (in & 1); return (int64_t)h * h + (int64_t)h * ip; }
  int in = gns_lin(d, h), ip = in + (in & 1), op = out + (out & 1);
  return (int64_t)op * h + (int64_t)h * h + (int64_t)h * ip;
}
GNS_HD static inline int64_t gns_n_block(bool is_phi, int d, int h, int out) {
  return gns_pad16(gns_n_w1x_off(is_phi, d, h, out) + (int64_t)gns_w1x_groups(is_phi, d, h) * h * 4);
}

struct GnsFamilies {   // per network family (phi*, L_theta, L_v, L_m) in state_dict order
  int nfam;            // 4 (single phi) or 6
  int nphi;            // 1 or 3
  int in[6], out[6];   // UNFOLDED shapes of the MODEL (the flat / state_dict layout)
  int phi_of[6];       // for an L family: the phi family whose message sum it reads (main.py:165-171)
  int64_t flat_off[6], t_off[6], n_off[6], g_off[6];   // offset of block k=0 of the family (g: folded-gradient slab layout)
  int64_t flat_sz[6], t_sz[6], n_sz[6], g_sz[6];       // per-k block size
  int64_t flat_total, t_total, n_total, g_total;
  // A model narrower than a compiled kernel runs on it ZERO-PADDED: hidden units with zero weights and biases stay 0 through
  // LeakyReLU and feed nothing, latent components beyond the model's start at 0 (main.py:141) and are updated by zero rows, and every
  // added term of every sum is an exact +0 - the same function and gradients, bit for bit what the kernel computes for the padded
  // model.  dr, hr = the model's (latent_dim, hidden_dim): the flat layout; dk, hk >= them = the kernel's: the t / n / g layouts.
  int dr, hr, dk, hk;
  int outk[6];         // output width of the family in kernel dims (dk for phi* of three-phi models and L_m, else 1)
};

static inline void gns_families_padded(int dr, int hr, int dk, int hk, int K, int multi, GnsFamilies* f) {
  f->nfam = multi ? 6 : 4;
  f->nphi = multi ? 3 : 1;
  f->dr = dr; f->hr = hr; f->dk = dk; f->hk = hk;
  int64_t fo = 0, to = 0, no = 0, go = 0;
  for (int i = 0; i < f->nfam; ++i) {
    bool is_phi = i < f->nphi;
    const bool wide_out = is_phi ? multi != 0 : i == f->nfam - 1;
    f->in[i] = is_phi ? dr + 5 : 4 + 2 * dr;
    f->out[i] = wide_out ? dr : 1;
    f->outk[i] = wide_out ? dk : 1;
    const int ink = is_phi ? dk + 5 : 4 + 2 * dk;
    // L_theta reads phi_theta, L_v reads phi_v, L_m reads phi_m; registration order is phi_v, phi_theta, phi_m (main.py:113-116)
    f->phi_of[i] = is_phi ? -1 : (multi ? (i == 3 ? 1 : (i == 4 ? 0 : 2)) : 0);
    f->flat_sz[i] = gns_flat_block(f->in[i], hr, f->out[i]);
    f->t_sz[i] = gns_t_block(is_phi, dk, hk, f->outk[i]);
    f->n_sz[i] = gns_n_block(is_phi, dk, hk, f->outk[i]);
    // gradient of the folded block: phi' W1[h][in] b1 W2 b2 ; L' W1'[h][in'] b1 W2 b2 W4[out][h] b4
    f->g_sz[i] = is_phi ? (int64_t)ink * hk + hk + (int64_t)hk * hk + hk
                        : (int64_t)gns_lin(dk, hk) * hk + hk + (int64_t)hk * hk + hk + (int64_t)f->outk[i] * hk + f->outk[i];
    f->flat_off[i] = fo; f->t_off[i] = to; f->n_off[i] = no; f->g_off[i] = go;
    fo += f->flat_sz[i] * K; to += f->t_sz[i] * K; no += f->n_sz[i] * K; go += f->g_sz[i] * K;
  }
  f->flat_total = fo; f->g_total = go; f->t_total = to + 64; f->n_total = no + 64;   // +64: the 16-float chunk loader may read past the end
}
static inline void gns_families(int d, int h, int K, int multi, GnsFamilies* f) { gns_families_padded(d, h, d, h, K, multi, f); }

// ---- forward workspace layout (byte offsets, 256-B aligned) -----------------------------------------
struct GnsFwdLayout {
  int64_t groups;        // ceil(Bt/64)
  int64_t mq;            // float4 rows holding the latent vector: ceil(d/4)
  int64_t rows_bus;      // 1 + mq
  int64_t slots;         // K+1 when the state is saved for backward, else 2
  size_t off_pt, off_pn, off_in, off_lam, off_state, off_msg, off_team, total;
};
#define GNS_TEAM_MAX_GROUPS 128   // teams only form when every group can have two CUs

static inline size_t gns_align256(size_t x) { return (x + 255) & ~(size_t)255; }

static inline void gns_fwd_layout(int N, int E, int d, int h, int K, int multi, int64_t Bt, int save, GnsFwdLayout* L) {
  GnsFamilies f; gns_families(d, h, K, multi, &f);
  L->groups = (Bt + GNS_LANES - 1) / GNS_LANES;
  L->mq = (d + 3) / 4;
  L->rows_bus = 1 + L->mq;
  L->slots = save ? K + 1 : 2;    // inference ping-pongs between two slots (families of one bus are updated by different waves)
  size_t o = 0;
  L->off_pt = o;    o = gns_align256(o + (size_t)f.t_total * 4);
  L->off_pn = o;    o = gns_align256(o + (size_t)f.n_total * 4);
  L->off_in = o;    o = gns_align256(o + (size_t)L->groups * gns_in_rows(N, E) * GNS_LANES * 16);
  L->off_lam = o;   o = gns_align256(o + (size_t)K * L->groups * GNS_LANES * 8);
  L->off_state = o; o = gns_align256(o + (size_t)L->slots * L->groups * N * L->rows_bus * GNS_LANES * 16);
  // hidden-vector sums per (step, bus, phi family): saved by the training forward so that the backward need not recompute them
  L->off_msg = o;   o = gns_align256(o + (save ? (size_t)K * L->groups * N * (multi ? 3 : 1) * ((h + 3) / 4) * GNS_LANES * 16 : 0));
  L->off_team = o;  o = gns_align256(o + (L->groups <= GNS_TEAM_MAX_GROUPS ? gns_team_bytes(L->groups) : 0));   // counters | partial sums
  L->total = o;
}

// ---- backward workspace layout ------------------------------------------------------------------------
#define GNS_BWD_WAVES 8        // waves per backward workgroup
#define GNS_RED_PARTS 64       // first-stage partial sums of the slab reduction
struct GnsBwdLayout {
  int64_t groups, mq, rows_bus;
  int64_t slab_floats;     // per-wave gradient slab: one float per FOLDED parameter
  int64_t adj_rows;        // adjoint rows per bus
  int64_t nslab;           // number of slabs (workgroups x waves)
  size_t off_adj, off_slots, off_slab, off_part, off_tmp, off_team, total;
};
#define GNS_BWD_MAX_WG 256   // persistent backward workgroups (each loops over grid groups)

static inline void gns_bwd_layout(int N, int E, int d, int h, int K, int multi, int64_t Bt, int team, GnsBwdLayout* B) {
  GnsFamilies f; gns_families(d, h, K, multi, &f);
  B->groups = (Bt + GNS_LANES - 1) / GNS_LANES;
  B->mq = (d + 3) / 4;
  B->rows_bus = 1 + B->mq;
  B->slab_floats = (f.g_total + 63) / 64 * 64;
  B->adj_rows = B->rows_bus + 1 + (multi ? 0 : (h + 3) / 4);   // (vbar,thbar,dpbar,-) | input adjoints | [hidden-sum adjoint, single phi] | mbar
  int64_t wg = B->groups * team < GNS_BWD_MAX_WG ? B->groups * team : GNS_BWD_MAX_WG;   // team > 1 only when groups * team fits
  B->nslab = wg * GNS_BWD_WAVES;
  size_t o = 0;
  B->off_adj = o;   o = gns_align256(o + (size_t)B->groups * N * B->adj_rows * GNS_LANES * 16);
  B->off_slots = o; o = gns_align256(o + (size_t)B->groups * 6 * E * GNS_LANES * 4);               // 6 adjoint planes per line
  B->off_slab = o;  o = gns_align256(o + (size_t)B->nslab * B->slab_floats * 4);
  B->off_part = o;  o = gns_align256(o + (size_t)GNS_RED_PARTS * B->slab_floats * 4);
  B->off_tmp = o;   o = gns_align256(o + (size_t)B->slab_floats * 4);
  B->off_team = o;  o = gns_align256(o + (team > 1 ? gns_team_bytes(B->groups) : 0));
  B->total = o;
}

// ---- split backward (bwd_variant 4): one kernel sequence per reverse step instead of one persistent kernel -----------------
// Per step k = K-1..0:  gns_bwds_phys_kernel (one 16-wave workgroup per 64-grid group: Pb-0, the lambda adjoint, the line
// phase and the per-bus gather) and gns_bwds_sweep_kernel (independent ONE-WAVE workgroups, one per (family, bus chunk,
// block of R groups); no barrier of any kind).  The three families of a step no longer add into one latent-adjoint row in
// turn: each writes its own part and the reader sums the parts in the old order, so they can run side by side and nothing
// has to meet at a counter in HBM - there are no teams in this variant.
//   adjoint rows per bus:  A0 (vbar, thbar, dpbar, 2 Gs v) | X[3] input-adjoint sums of L_theta, L_v, L_m |
//                          M[2 step parities][3 families][mq] parts of the latent adjoint
#define GNS_BWDS_PHYS_WAVES 16
struct GnsBwdsLayout {
  int64_t groups, mq, adj_rows, slab_floats;
  int C;                   // bus chunks per group (partition table for C waves)
  int R;                   // groups per sweep workgroup (accumulator tiles stay in registers across them)
  int64_t gblocks;         // ceil(groups / R)
  int64_t nslab;           // gblocks * C, one slab per (group block, chunk); every (family, step) block is stored exactly once
  size_t off_adj, off_slots, off_slab, off_part, off_tmp, total;
};
static inline void gns_bwds_layout(int N, int E, int d, int h, int K, int multi, int64_t Bt, int ncu, int chunks, GnsBwdsLayout* B) {
  GnsFamilies f; gns_families(d, h, K, multi, &f);
  B->groups = (Bt + GNS_LANES - 1) / GNS_LANES;
  B->mq = (d + 3) / 4;
  B->adj_rows = 4 + 6 * B->mq;
  B->slab_floats = (f.g_total + 63) / 64 * 64;
  if (ncu <= 0) ncu = 256;
  // a sweep kernel should put 8 one-wave workgroups on every CU (two per SIMD: more were measured slower, the sweeps are bound
  // by the rows they stream): 8 bus chunks per group when there is a group per CU, finer chunks for smaller batches, several
  // groups per workgroup for larger ones
  const int64_t slots = 8LL * ncu;
  B->C = B->groups * 8 >= slots ? 8 : (B->groups * 16 >= slots ? 16 : 32);
  B->R = (int)((B->groups * B->C) / slots);
  if (B->R < 1) B->R = 1;
  if (chunks > 0 && gns_part_index(chunks) >= 0) B->C = chunks;             // explicit ("bwds_chunks")
  B->R = (int)((B->groups * B->C) / slots);
  if (B->R < 1) B->R = 1;
  B->gblocks = (B->groups + B->R - 1) / B->R;
  B->nslab = B->gblocks * B->C;
  size_t o = 0;
  B->off_adj = o;   o = gns_align256(o + (size_t)B->groups * N * B->adj_rows * GNS_LANES * 16);
  B->off_slots = o; o = gns_align256(o + (size_t)B->groups * 6 * E * GNS_LANES * 4);
  B->off_slab = o;  o = gns_align256(o + (size_t)B->nslab * B->slab_floats * 4);
  B->off_part = o;  o = gns_align256(o + (size_t)GNS_RED_PARTS * B->slab_floats * 4);
  B->off_tmp = o;   o = gns_align256(o + (size_t)B->slab_floats * 4);
  B->total = o;
}
