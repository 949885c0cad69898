/* C-ABI of the MI355X-native GNS K-step hot path (libgns_hip.so).
 *
 * Drop-in boundary for the reference's GNS.forward / autograd backward
 * (LeonOrou/OPF-Graph-Neural-Solver, GNS/main.py:140-202 and the .backward() of main.py:288).
 * Plain pointers and sizes only; every device pointer is owned by the caller (torch), nothing is
 * allocated, synchronised or copied host<->device inside gns_forward / gns_backward, and all work is
 * enqueued on the caller's stream, so the calls are hipGraph-capturable.  Return 0 = ok, otherwise a
 * GNS_E* code; nothing throws across this boundary.
 *
 * Layouts
 *   params / grad_params : ONE flat fp32 buffer in the reference's state_dict order
 *                          (GNS/main.py:113-134: {phi | phi_v,phi_theta,phi_m}.k, L_theta.k, L_v.k, L_m.k;
 *                          per block linear1.weight[h,in], linear1.bias, linear2.*, linear4.*), nn.Linear
 *                          [out,in] row-major.  The same buffer is the RCCL all-reduce message.
 *   buses [Bt,N,6], lines [Bt,E,7], generators [Bt,Gn,7] : the reference's column order
 *                          (GNS/utils.py:4-13), fp32, contiguous - what utils.load_all_grids returns.
 *   v, theta [Bt,N]; total_loss, last_loss [Bt].
 *   Topology (f_bus, t_bus, generator buses) is shared by the whole batch (reference data:
 *   GNS/augment_grids.py:35-53 perturbs continuous columns only).
 */
#ifndef GNS_HIP_H
#define GNS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  GNS_OK = 0,
  GNS_EINVAL = 1,      /* null pointer / non-positive size / malformed config            */
  GNS_EUNSUPPORTED = 2,/* no compiled kernel holds this (latent_dim, hidden_dim), or K > 64 */
  GNS_ETOPOLOGY = 3,   /* bus id out of range, or a bus id that is not a valid line index
                          (the reference gathers per-line arrays with bus ids, main.py:41)  */
  GNS_ESIZE = 4,       /* caller-provided buffer too small                               */
  GNS_ELAUNCH = 5      /* HIP reported a launch error                                    */
};

/* Constructor arguments of the reference's GNS (GNS/main.py:108) plus the grid shape. */
typedef struct gns_config {
  int32_t n_bus;        /* N  */
  int32_t n_line;       /* E  */
  int32_t n_gen;        /* Gn */
  int32_t K;            /* correction steps                    (main.py:138) */
  int32_t latent_dim;   /* d                                   (main.py:136) */
  int32_t hidden_dim;   /* h                                   (main.py:18)  */
  int32_t multiple_phi; /* 0: one phi net, 1: phi_v/theta/m    (main.py:111) */
  float   gamma;        /* loss discount                       (main.py:137) */
} gns_config;

/* Library / build identification: returns a static string such as "gns_hip 0.1 gfx950". */
const char* gns_version(void);

/* Number of fp32 parameters of GNS(latent_dim, hidden_dim, K, multiple_phi) = length of the flat
 * params buffer (GNS/main.py:113-134).  */
int gns_param_count(const gns_config* cfg, int64_t* count);

/* 1 if a compiled kernel holds this model, else 0.  Kernels are compiled for (latent_dim, hidden_dim) = (20, 10), (10, 10) and (20, 14),
 * both phi modes; a narrower model (any latent_dim <= 20, hidden_dim <= 14: the "more or less hidden dim" of main.py:215) runs on the
 * smallest pair that holds it zero-padded - same outputs, same gradients, delivered in the MODEL's flat layout (gns_param_count floats). */
int gns_config_supported(const gns_config* cfg);

/* Host-side topology preparation (replaces the per-call index construction of main.py:35-36,85-86,144,153):
 * builds, from 0-based src/dst/gen_bus HOST arrays, the destination-sorted and source-sorted CSR edge lists,
 * the bus-id-as-line-index tuples, the adjoint incidence lists and the per-wave bus partition, into a
 * relocatable blob of gns_topology_bytes() bytes that the caller copies to the device once per case. */
int gns_topology_bytes(int32_t n_bus, int32_t n_line, int32_t n_gen, size_t* bytes);
int gns_prepare_topology(int32_t n_bus, int32_t n_line, int32_t n_gen,
                         const int32_t* src, const int32_t* dst, const int32_t* gen_bus,
                         void* topo_host_out, size_t topo_bytes);

/* Device workspace sizes for a batch of Bt grids.
 *   fwd_bytes       : forward workspace; with save_state != 0 it also holds the K+1 per-step states the
 *                     backward pass re-reads, and must stay untouched until gns_backward has run.
 *   bwd_bytes       : extra scratch for gns_backward (adjoint state + per-wave gradient slabs).  */
int gns_workspace_bytes(const gns_config* cfg, int64_t Bt, int save_state, size_t* fwd_bytes, size_t* bwd_bytes);

/* A batch that stays resident across many steps (an epoch over device-resident data, the benchmark's batch) can be brought
 * into the lane-per-grid kernels' input layout ONCE: gns_prepack writes gns_prepack_bytes() bytes that gns_forward /
 * gns_backward then read through `packed_inputs` instead of re-packing on every call (the batch-invariant prologue of
 * main.py:144-152 and the bus-id-as-line-index gathers of main.py:38,41).  Pass NULL to pack per call.  The buffer is only
 * valid for the tensors it was made from; the grid-per-workgroup kernels read the caller's tensors directly and ignore it. */
int gns_uses_packed_inputs(const gns_config* cfg, int64_t Bt, int save_state);   /* 1: the call would run the kernels that read `packed_inputs` */
int gns_prepack_bytes(const gns_config* cfg, int64_t Bt, size_t* bytes);
int gns_prepack(const gns_config* cfg, const void* topo_dev, const float* buses, const float* lines,
                const float* generators, int64_t Bt, void* packed, size_t packed_bytes, void* stream);

/* GNS.forward for Bt grids (GNS/main.py:140-202).  v/theta/total_loss/last_loss are written.
 * stream is a hipStream_t passed as void*.  */
int gns_forward(const gns_config* cfg, const void* topo_dev, const float* params,
                const float* buses, const float* lines, const float* generators, int64_t Bt,
                const void* packed_inputs,
                float* v, float* theta, float* total_loss, float* last_loss,
                void* workspace, size_t workspace_bytes, int save_state, void* stream);

/* Reverse pass of the same graph (what total_loss.backward() does at GNS/main.py:288).
 * grad_total / grad_last [Bt] and grad_v / grad_theta [Bt,N] are upstream gradients (any may be NULL = 0).
 * grad_params (flat, state_dict order) is ACCUMULATED into (+=), like autograd does with .grad.
 * L_m.{K-1} (and phi_m.{K-1}) receive exactly zero, matching the reference where they get no gradient.
 * buses / lines / generators are the tensors the matching gns_forward was given (still owned by the caller, unchanged):
 * the grid-per-workgroup kernels re-read the batch-constant inputs from them; the lane-per-grid kernels ignore them.
 * The mapping ("train_mapping", "gw_pack" options) must not change between a forward and its backward. */
int gns_backward(const gns_config* cfg, const void* topo_dev, const float* params,
                 const float* buses, const float* lines, const float* generators, int64_t Bt,
                 const void* packed_inputs,
                 const void* fwd_workspace, size_t fwd_workspace_bytes,
                 const float* grad_total, const float* grad_last, const float* grad_v, const float* grad_theta,
                 float* grad_params, void* bwd_workspace, size_t bwd_workspace_bytes, void* stream);

/* Teams (lane-per-grid kernels, batches with fewer 64-grid groups than CUs: "team" option below) meet at counters in the workspace.
 * A barrier whose partner workgroup never becomes resident - another kernel or process holds its CU - gives up after ~seconds; the
 * losses of that call are NaN and the workspace's status word is set.  gns_team_status reads that word: *status = 1 if a team gave
 * up during the gns_forward that used `fwd_workspace` (same cfg, Bt, save_state), else 0.  It is the ONE entry point that
 * synchronises (it waits for `stream`, then copies one word); it returns 0 without touching the device when the call did not use
 * teams.  The host wrapper asks before it launches a backward, so that no gradient of invalid losses reaches an optimiser. */
int gns_team_status(const gns_config* cfg, int64_t Bt, const void* fwd_workspace, size_t fwd_workspace_bytes, int save_state,
                    int* status, void* stream);
/* Byte offset of the status word inside the forward workspace; (size_t)-1 when this (cfg, Bt, save_state) runs without teams, i.e.
 * when there is nothing to check.  Host code only, no device access. */
int gns_team_status_offset(const gns_config* cfg, int64_t Bt, int save_state, size_t* offset);

/* The optimiser update of the reference's training loop (optimizer.step() at GNS/main.py:290 with torch.optim.Adam,
 * main.py:241-243: no weight decay, no amsgrad) on the ONE flat parameter buffer, in one launch:
 *   m = m + (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;
 *   p -= (lr / (1 - beta1^step)) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)          (step counts from 1)
 * params / grad / exp_avg / exp_avg_sq: n floats each on the device, caller-owned; grad is read only.  The hyper-parameters are
 * doubles like torch's (1 - beta is formed in double, then rounded to the float the kernel multiplies with). */
int gns_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  double lr, double beta1, double beta2, double eps, int64_t step, void* stream);

/* gns_adam_step with the step counter in device memory - for a training step captured into a HIP graph (a captured launch bakes
 * its scalar arguments in; Adam's bias corrections change every step).  step_state: 4 floats on the device, caller-owned, zero
 * before the first step: [0] steps taken so far, [1..2] scratch of the current step.  Every call advances [0] by one.  Two
 * launches on `stream`, no synchronisation: capturable.  The same arithmetic as gns_adam_step (corrections formed in double). */
int gns_adam_step_dev(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                      double lr, double beta1, double beta2, double eps, float* step_state, void* stream);

/* ---- diagnostics (benchmarks only; not on the reference's interface) -------------------------------
 * gns_profile_enable(capacity > 0): from now on gns_forward / gns_backward record a HIP event pair around
 * their fused main kernel, on the caller's stream, into a ring of `capacity` pairs per direction (capacity 0
 * switches it off and frees the events).  gns_profile_read waits for the recorded events, returns the summed
 * kernel time in milliseconds and the number of launches since the last read, and rewinds the ring.
 * Not graph-capturable and not thread-safe; leave it off in production. */
int gns_profile_enable(int capacity);
int gns_profile_read(int backward, float* ms_sum, int* launches);

/* ---- configuration (process-wide; set before launching, not concurrently with launches) ------------
 * The library reads its environment ONCE, at the first call of any entry point below or of gns_forward /
 * gns_backward (GNS_FWD_MAPPING=lane|lds, GNS_TRAIN_MAPPING=lane|lds, GNS_GW_PACK, GNS_FWD_WAVES, GNS_FWD_PLANE, GNS_DW_MFMA); these
 * setters override those defaults explicitly.  Options:
 *   "fwd_mapping" evaluation-mode forward: 0 auto | 1 lane-per-grid kernels (state streamed through HBM) | 2 grid-per-workgroup kernels (state on chip)
 *   "train_mapping" training-mode forward + backward pair: same values
 *   "gw_pack"     grids per workgroup of the grid-per-workgroup mapping (0 = auto)
 *   "fwd_waves"   waves per workgroup of the lane-per-grid forward (1,2,4,8,16)
 *   "fwd_plane"   LDS planes of the lane-per-grid forward: 0 none (neighbour (v, theta) gathered from HBM) | 1 the (v, theta) plane |
 *                 2 (default) also (delta_p, delta_q) between the physics and the lambda phase of a step
 *   "bwd_variant" lane-per-grid backward: one persistent kernel with 1 wide half-wave records | 2 layer-wise sweep with sub-record
 *                 windows | 3 = 2 with the contraction chains issued behind the weight streams; 4 (default, three-phi models on the
 *                 matrix pipe) one kernel sequence per reverse step: no teams, no spin-waits (gns_backward_split.hip)
 *   "bwds_mode"   sweep kernels per reverse step of variant 4: 0 one per family | 1 (default) {L_m} {L_theta + L_v} | 2 all three
 *                 families of a bus in one kernel
 *   "bwds_chunks" bus chunks per 64-grid group of variant 4's sweeps: 0 auto | 8 | 12 | 16 | 24 | 32
 *   "dw_mfma"     0: weight-gradient contraction on packed FMAs instead of the fp32 matrix pipe
 *   "team"        lane-per-grid kernels, batches with fewer 64-grid groups than CUs: workgroups per group, 0 auto (as many
 *                 as keep every workgroup resident by the kernel's own occupancy at that launch configuration) | 1 none | 2 | 4.
 *                 Teams meet at counters in the workspace and need the device to themselves: a kernel of another stream or
 *                 process holding a partner's CU makes a barrier give up (bounded: NaN losses + gns_team_status, never a hang).
 *                 Only the forward uses them by default (the default backward, variant 4, has none).
 * Both mappings and both engines compute the same function of the reference (GNS/main.py:140-202, :288). */
int gns_set_option(const char* name, int value);
int gns_get_option(const char* name, int* value);

#ifdef __cplusplus
}
#endif
#endif /* GNS_HIP_H */
