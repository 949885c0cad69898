// Kernel argument blocks and launch entry points shared between the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/gns_hip.h"
#include "gns_common.h"

#define GNS_FWD_THREADS 1024     // default: 16 waves (4 per SIMD) split the buses of 64 grids; GNS_FWD_WAVES=8 selects 8
#define GNS_FWD_MAX_THREADS 1024 // register budget of the forward kernel: 128 VGPRs -> 4 waves/SIMD
#define GNS_BWD_THREADS 512
#define GNS_MAX_K 64
// forward (lane-per-grid): dynamic LDS = the (v, theta) plane [N][64] float2 (when it fits) + the per-wave partial sums
// red[2][GNS_MAXW][64][2] (a team keeps those in HBM, which is what lets the case300 plane fit); larger cases read
// neighbours from HBM.  Static LDS: the unit counters and the team flag.
#define GNS_FWD_RED_BYTES (2 * GNS_MAXW * GNS_LANES * 2 * 4)
#define GNS_FWD_DYN_LDS_MAX (160 * 1024 - 64)
static inline int gns_fwd_plane_fits(int N, int team) {
  return (size_t)N * GNS_LANES * 8 + (team == 1 ? GNS_FWD_RED_BYTES : 0) <= (size_t)GNS_FWD_DYN_LDS_MAX;
}
// ... and the second plane (delta_p, delta_q between the physics and the lambda phase): one workgroup per group only
static inline int gns_fwd_plane2_fits(int N, int team) {
  return team == 1 && 2 * (size_t)N * GNS_LANES * 8 + GNS_FWD_RED_BYTES <= (size_t)GNS_FWD_DYN_LDS_MAX;
}

// (latent_dim, hidden_dim) pairs with compiled kernels (a narrower model runs zero-padded on the smallest pair that holds it)
#define GNS_FOR_EACH_DIMS(X) X(20, 14) X(20, 10) X(10, 10)
// ... of the persistent lane-per-grid backward kernels (bwd_variant 1-3, the packed-FMA engine): their record windows are laid out for
// hidden_dim 10; wider pairs always take the split backward or the grid-per-workgroup pair
#define GNS_FOR_EACH_DIMS_PERSISTENT(X) X(20, 10) X(10, 10)
// ... and of the grid-per-workgroup backward (its staged windows too): a (20, 14) model evaluates on either mapping and trains on the
// lane-per-grid forward + split backward
#define GNS_FOR_EACH_DIMS_GWB(X) X(20, 10) X(10, 10)

template <int D, int H, bool MULTI>
struct GnsDims {
  static constexpr int NPHI = MULTI ? 3 : 1;
  static constexpr int PHI_IN = D + 5;                 // [m(dst) | r x b tau shift]        main.py:155
  static constexpr int PHI_OUT = MULTI ? D : 1;        // main.py:126-130
  static constexpr int PHI_OUTP = PHI_OUT + (PHI_OUT & 1);
  static constexpr int L_IN = 4 + 2 * D;               // [v theta dp dq | m | phi_sum]      main.py:165-171 (flat layout)
  static constexpr int LF_IN = 4 + D + H + 1;          // folded: [v theta dp dq | m | sum_e h_e | deg]
  static constexpr int HQ = (H + 3) / 4;               // float4 rows of a hidden vector
  static constexpr int MQ = (D + 3) / 4;               // float4 rows of the latent vector
  static constexpr int RB = 1 + MQ;                    // state rows per bus: (v,theta,dp,dq) + m
};

struct GnsFwdArgs {
  const int* topo;
  const float* pt;        // T-stream parameters
  const float* in;        // packed inputs
  float* state;           // [slots][G][N][RB][64] float4
  float* lam;             // [K][G][64] float2 (lambda, branch bits) when save != 0
  float* msg;             // [K][G][N][NPHI][HQ][64] float4: hidden-vector sums, when save != 0
  float* v_out; float* theta_out; float* total_out; float* last_out;
  long long t_off[6], t_sz[6];
  float gw[GNS_MAX_K];    // gamma^(K-k) rounded to fp32 from a double, like the reference's python float
  long long Bt, G;
  int N, E, K, save, part_idx;
  int plane;              // 1: the (v, theta) of the step being produced is mirrored in LDS ([N][64] float2, dynamic shared memory); 2: and (delta_p, delta_q) between the physics and lambda phases
  int team;               // workgroups per 64-grid group (1 = none); part_idx then names the partition for team * waves
  unsigned char* team_ws; // team > 1: [G] 64-byte counter lines (zero at launch) | [G][GNS_TEAM_RED_FLOATS] partial sums
};

struct GnsBwdArgs {
  const int* topo;
  const float* pt; const float* pn;      // T-stream (recompute) and N-stream (data gradients) parameters
  const float* in;                       // packed inputs
  const float* state;                    // saved states S_0..S_K of the forward
  const float* lam;                      // (lambda, branch bits) per step
  const float* msg;                      // hidden-vector sums per (step, bus, phi family) saved by the forward
  const float* g_total; const float* g_last; const float* g_v; const float* g_theta;   // upstream gradients (nullable)
  float* adj;                            // [G][N][RB][64] float4: (vbar, thbar, dpbar, -) + mbar
  float* slots;                          // [G][6][E][64] per-line physics adjoints
  float* slab;                           // [blocks*8][slab_floats] per-wave weight-gradient accumulators
  long long t_off[6], t_sz[6], n_off[6], n_sz[6], g_off[6], g_sz[6];
  float gw[GNS_MAX_K];
  long long Bt, G, slab_floats;
  int N, E, K, part_idx;
  int team; unsigned char* team_ws;      // as in GnsFwdArgs; blocks = teams * team, slabs per (block, wave)
  int slab_dirty;                        // 1: the slabs were NOT zeroed by the caller; the V2 sweep stores (instead of adding) on a workgroup's first group
};

int gns_launch_backward(int d, int h, int multi, int mfma, int variant, const GnsBwdArgs& A, int blocks, hipStream_t st);
int gns_backward_persistent_supported(int d, int h);
int gns_launch_reduce(const float* slab, float* part, float* tmp, const float* flat, float* grad, long long nslab, long long sf,
                      const GnsFamilies& fam, int K, int D, int H, hipStream_t st, long long stride = 0);   // stride (floats) between the slabs read; 0: sf
int gns_launch_forward(int d, int h, int multi, const GnsFwdArgs& A, int threads, hipStream_t st);
int gns_fwd_init_device();
int gns_fwd_blocks_per_cu(int d, int h, int multi, const GnsFwdArgs& A, int threads);
int gns_launch_pack_params(const float* flat, float* pt, float* pn, const GnsFamilies& fam, int K, int D, int H, hipStream_t st);
int gns_launch_pack_inputs(const int* topo, const float* buses, const float* lines, const float* gens, float* out, int N,
                           int E, int Gn, long long Bt, long long groups, hipStream_t st);

// split backward (gns_backward_split.hip; gns_common.h "split backward")
struct GnsBwdsArgs {
  const int* topo;
  const float* pt; const float* pn;
  const float* in; const float* state; const float* lam; const float* msg;
  const float* g_total; const float* g_last; const float* g_v; const float* g_theta;
  float* adj; float* slots; float* slab;
  long long t_off[6], t_sz[6], n_off[6], n_sz[6], g_off[6], g_sz[6];
  long long Bt, G, slab_floats;
  float gwk;               // gamma^(K-k) of the step being reversed
  int N, E, K, k;
  int C, part_idx, R;      // sweep: bus chunks per group, their partition table, groups per workgroup
  int RB, RBA;             // state rows per bus (1 + mq), adjoint rows per bus (4 + 6 mq)
  int use_plane;           // phys: LDS planes of the line phase: 2 (v, theta, dpbar) of all buses, 1 (v, theta) only, 0 none
  int mode;                // sweep kernels per step: 0 {m}{theta}{v}, 1 {m}{theta+v}, 2 {m+theta+v} (gns_backward_split.hip)
};
int gns_bwds_supported(int d, int h, int multi);
size_t gns_bwds_phys_lds(int N, int* use_plane);
int gns_launch_bwds_phys(const GnsBwdsArgs& A, size_t lds, hipStream_t st);
int gns_launch_bwds_sweep(int d, int h, int multi, const GnsBwdsArgs& A, hipStream_t st);
int gns_bwds_init_device();
