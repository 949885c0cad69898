// Grid-per-workgroup ("gw") mapping of the GNS hot path: argument blocks, LDS layout, launch entry points.
//
// One grid (or a pack of P grids) lives in ONE workgroup: a bus lane owns one bus of one grid and keeps its state
// (v, theta, delta_p, delta_q, latent vector) in REGISTERS for all K steps, an edge lane owns one line.  Everything
// that crosses lanes goes through LDS (message sums, the (v, theta) plane of the line physics, per-line physics
// terms); HBM sees the compulsory bytes only (inputs once, outputs once, and in training mode the per-step saves the
// backward needs).  The MLP weights stay wave-uniform - every bus / line of a step uses the same LearningBlock - so
// they still travel through the scalar unit exactly as in the lane-per-grid kernels (gns_device.h).
#pragma once
#include "gns_kernels.h"

// LDS image of one grid slot, in floats (every region starts on a 16-byte boundary)
struct GwLds { int u, h, plane, phys, red, total; };
#define GW_RED_FLOATS(wpg) (12 * (wpg))   // [2 parities][2 kinds][wpg] + gsum [4][wpg] + epilogue [2][wpg] + pad
GNS_HD static inline GwLds gw_lds_layout(int N, int E, int UW, int WPG) {
  GwLds L; int o = 0;
  L.u = o;     o += (N * UW + 3) & ~3;        // head of phi' per bus: [N][UW]           bus -> edge
  L.h = o;     o += (E * UW + 3) & ~3;        // hidden vector of phi' per line: [E][UW]   edge -> bus
  L.plane = o; o += (2 * N + 3) & ~3;         // (v, theta) of the step being produced     bus -> edge
  L.phys = o;  o += 4 * E;                    // (p_from, q_from, p_to, q_to) per line     edge -> bus
  L.red = o;   o += (GW_RED_FLOATS(WPG) + 3) & ~3;
  L.total = o;
  return L;
}

struct GnsGwFwdArgs {
  const int* topo;
  const float* pt;                       // T-stream parameters (gns_pack_params_kernel)
  const float* buses; const float* lines; const float* gens;   // the caller's tensors, reference layout
  float* v_out; float* theta_out; float* total_out; float* last_out;
  float* sv_state;                       // save != 0: [K][Bt][SVQ][N] float4: state entering step k, bus-lane order
  float* sv_S;                           //            [K][Bt][SSQ][N] float4: hidden-vector sums per phi family
  float* sv_lam;                         //            [K][Bt] float2 (lambda, branch bits)
  long long t_off[6], t_sz[6];
  float gw[GNS_MAX_K];
  long long Bt;
  int N, E, Gn, K, save;
  int P, WPG;                            // grids per workgroup, waves per grid
};

// ---- saved quantities of a training-mode forward (what the backward re-reads), all in bus-lane order -------------------
//   state [K+1][Bt][SVQ][N] float4 : (v, theta, dp, dq) + latent vector ENTERING step k; slot K holds row 0 only
//   S     [K][Bt][NPHI*HQ][N] float4: per phi family the summed hidden vector of the lines ending at the bus
//   lam   [K][Bt] float2           : (lambda, branch bits)
struct GwSaveLayout { size_t off_state, off_S, off_lam, total; };
static inline GwSaveLayout gw_save_layout(int N, int d, int h, int K, int multi, int64_t Bt) {
  GwSaveLayout L; size_t o = 0;
  const size_t svq = 1 + (d + 3) / 4, ssq = (size_t)(multi ? 3 : 1) * ((h + 3) / 4);
  L.off_state = o; o = gns_align256(o + (size_t)(K + 1) * Bt * svq * N * 16);
  L.off_S = o;     o = gns_align256(o + (size_t)K * Bt * ssq * N * 16);
  L.off_lam = o;   o = gns_align256(o + (size_t)K * Bt * 8);
  L.total = o;
  return L;
}

struct GwBwdLds { int plane3, slots, gS, u, g1, red, ylds, rec, stage, total; };
GNS_HD static inline GwBwdLds gw_bwd_lds_layout(int N, int E, int H, int WPG, int recf, int stgf) {
  // Two regions are shared in time: the per-line physics adjoints (P1 -> P2) live where the first-layer adjoints of phi'
  // (E -> B') go later in the step, and the (v, theta, dpbar) plane (P0 -> P1) where the hidden-sum adjoints (B -> E) go.
  GwBwdLds L; int o = 0;
  const int g1f = E * H > 6 * E ? E * H : 6 * E, gSf = N * H > 3 * N ? N * H : 3 * N;
  L.gS = o; L.plane3 = o; o += (gSf + 3) & ~3;   // bus -> edge
  L.u = o;      o += (N * H + 3) & ~3;           // bus share of phi' (one family), recomputed         bus -> edge
  L.g1 = o; L.slots = o; o += (g1f + 3) & ~3;    // edge -> bus
  L.red = o;    o += (8 * WPG + 3) & ~3;         // [2 parities][wpg] lambda-adjoint partials + gsum [4][wpg] + pad
  L.ylds = o;   o += (E + 3) & ~3;               // y = 1/sqrt(r^2+x^2) per line number, once per grid
  L.rec = o;    o += WPG * recf;                 // per-wave sub-record window of the weight-gradient contraction
  L.stage = o;  o += WPG * stgf;                 // per-wave stage of one gradient block between the matrix pipe and the slab
  L.total = o;
  return L;
}

struct GnsGwBwdArgs {
  const int* topo;
  const float* pt; const float* pn;
  const float* buses; const float* lines; const float* gens;
  const float* sv_state; const float* sv_S; const float* sv_lam;
  const float* g_total; const float* g_last; const float* g_v; const float* g_theta;   // upstream gradients (nullable)
  float* slab;                           // [blocks * waves][slab_floats] running weight-gradient sums (zeroed by the caller)
  long long t_off[6], t_sz[6], n_off[6], n_sz[6], g_off[6], g_sz[6];
  float gw[GNS_MAX_K];
  long long Bt, slab_floats;
  int N, E, Gn, K;
  int P, WPG;
};

int gns_gw_launch_forward(int d, int h, int multi, const GnsGwFwdArgs& A, hipStream_t st);
int gns_gw_launch_backward(int d, int h, int multi, const GnsGwBwdArgs& A, int blocks, hipStream_t st);
int gns_gw_backward_blocks(int N, int E, int d, int h, int multi, int P, long long Bt);   // persistent grid size (also the slab count / waves)
int gns_gw_backward_supported(int N, int E, int d, int h, int multi, int P);
int gns_gw_backward_wpg(int N);             // waves per grid of the backward: ceil(N / 64); a lane takes up to two lines
// 1 when the mapping can run this shape (LDS image fits, waves per workgroup <= 16)
int gns_gw_supported(int N, int E, int d, int h, int multi, int P);
int gns_gw_backward_init_device(void);
int gns_gw_init_device(void);            // one-time per process: opt-in to > 64 KB of dynamic LDS for every instantiation
