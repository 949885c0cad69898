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

int gns_gw_launch_forward(int d, int h, int multi, const GnsGwFwdArgs& A, hipStream_t st);
// 1 when the mapping can run this shape (LDS image fits, waves per workgroup <= 16)
int gns_gw_supported(int N, int E, int d, int h, int multi, int P);
int gns_gw_init_device(void);            // one-time per process: opt-in to > 64 KB of dynamic LDS for every instantiation
