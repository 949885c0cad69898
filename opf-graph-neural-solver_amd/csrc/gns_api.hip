// C-ABI entry points (include/gns_hip.h).  Host code only: validates, lays out the workspace and enqueues
// kernels on the caller's stream.  No allocation, no synchronisation, no host<->device copies.
#include <atomic>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include "gns_kernels.h"
#include "gns_gridwg.h"

// ---- process-wide tuning knobs: read from the environment ONCE (first call), never per launch ----------------------
namespace {
struct GnsTuning {
  int fwd_mapping;   // GNS_FWD_MAPPING: 0 auto, 1 "lane" (lane = grid, state streamed through HBM), 2 "lds" (grid per workgroup, state on chip)
  int gw_pack;       // GNS_GW_PACK: grids per workgroup of the lds mapping (0 = auto)
  int fwd_waves;     // GNS_FWD_WAVES: waves per workgroup of the lane mapping
  int fwd_plane;     // GNS_FWD_PLANE: LDS planes of the lane-mapping forward: 0 none (neighbour (v, theta) from HBM), 1 the (v, theta) plane, 2 (default) also (delta_p, delta_q) between the physics and lambda phases
  int dw_mfma;       // GNS_DW_MFMA=0: packed-FMA weight-gradient engine instead of the matrix pipe
  int train_mapping; // GNS_TRAIN_MAPPING: mapping of the training-mode forward + backward pair: 0 auto, 1 lane, 2 lds
  int bwd_variant;   // GNS_BWD_VARIANT: lane-per-grid backward: 1 wide half-wave records, 2 layer-wise + sub-record windows, 3 = 2 + background chains (one persistent kernel each); 4 split: one kernel sequence per reverse step (gns_backward_split.hip)
  int team;          // GNS_TEAM: workgroups per 64-grid group of the lane mapping when the batch leaves CUs idle: 0 auto, 1 none, 2, 4
  int bwds_mode;     // GNS_BWDS_MODE: sweep kernels per reverse step of the split backward: 0 one per family, 1 {L_m} {L_theta + L_v}, 2 all three families per bus in one kernel
  int bwds_chunks;   // GNS_BWDS_CHUNKS: bus chunks per 64-grid group of the split backward's sweeps (0 = auto: 12, 24 or 32)
};
GnsTuning make_tuning() {
  GnsTuning t{0, 0, GNS_FWD_THREADS / 64, 2, 1, 0, 4, 0, 1, 0};
  if (const char* e = std::getenv("GNS_TEAM")) { const int v = std::atoi(e); if (v >= 0 && v <= GNS_MAX_TEAM) t.team = v; }
  if (const char* e = std::getenv("GNS_FWD_MAPPING")) t.fwd_mapping = !std::strcmp(e, "lane") ? 1 : (!std::strcmp(e, "lds") ? 2 : 0);
  if (const char* e = std::getenv("GNS_GW_PACK")) { const int p = std::atoi(e); if (p >= 1 && p <= 16) t.gw_pack = p; }
  if (const char* e = std::getenv("GNS_FWD_WAVES")) { const int w = std::atoi(e); if (w > 0 && (w & (w - 1)) == 0 && w * 64 <= GNS_FWD_MAX_THREADS) t.fwd_waves = w; }
  if (const char* e = std::getenv("GNS_FWD_PLANE")) t.fwd_plane = e[0] == '0' ? 0 : (e[0] == '1' ? 1 : 2);
  if (const char* e = std::getenv("GNS_DW_MFMA")) t.dw_mfma = e[0] == '0' ? 0 : 1;
  if (const char* e = std::getenv("GNS_BWD_VARIANT")) { const int v = std::atoi(e); if (v >= 1 && v <= 4) t.bwd_variant = v; }
  if (const char* e = std::getenv("GNS_BWDS_MODE")) { const int v = std::atoi(e); if (v >= 0 && v <= 2) t.bwds_mode = v; }
  if (const char* e = std::getenv("GNS_BWDS_CHUNKS")) { const int v = std::atoi(e); if (v == 0 || gns_part_index(v) >= 0) t.bwds_chunks = v; }
  if (const char* e = std::getenv("GNS_TRAIN_MAPPING")) t.train_mapping = !std::strcmp(e, "lane") ? 1 : (!std::strcmp(e, "lds") ? 2 : 0);
  return t;
}
GnsTuning& tuning() {
  static GnsTuning t = make_tuning();            // C++11: thread-safe one-time initialisation
  return t;
}

// ---- per-device state: kernel attributes (dynamic LDS beyond 64 KB is an opt-in per kernel AND device) and the CU count, set up
// once per device at the first call that finds it current (one process per GPU is the normal deployment; a process that moves a
// model to a second device gets that device initialised the same way).  Never runs inside a stream capture in practice: a capture
// is preceded by eager warm-up calls on the same device.
constexpr int GNS_MAX_DEVICES = 64;
struct GnsDevice {
  int ncu;           // compute units (teams must be resident all at once)
  int gw_ready;      // gns_gw_init_device() and gns_gw_backward_init_device() succeeded
  int split_ready;   // gns_bwds_init_device() succeeded
  int fwd_ready;     // gns_fwd_init_device() succeeded (the lane-per-grid forward may use more than 64 KB of dynamic LDS)
};
const GnsDevice& device() {
  static GnsDevice devs[GNS_MAX_DEVICES];
  static std::atomic<int> done[GNS_MAX_DEVICES];
  static std::mutex mu;
  static const GnsDevice none{0, 0, 0, 0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return none; }     // no ROCm device: the entry points report it
  if (dev < 0 || dev >= GNS_MAX_DEVICES) return none;
  if (!done[dev].load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lock(mu);
    if (!done[dev].load(std::memory_order_relaxed)) {
      GnsDevice d{0, 0, 0, 0};
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) d.ncu = n;
      else (void)hipGetLastError();
      d.gw_ready = (gns_gw_init_device() == GNS_OK && gns_gw_backward_init_device() == GNS_OK) ? 1 : 0;
      d.split_ready = gns_bwds_init_device() == GNS_OK ? 1 : 0;
      d.fwd_ready = gns_fwd_init_device() == GNS_OK ? 1 : 0;
      devs[dev] = d;
      done[dev].store(1, std::memory_order_release);
    }
  }
  return devs[dev];
}
}  // namespace

// Explicit configuration (include/gns_hip.h).  The environment variables of the same meaning only seed the defaults.
extern "C" int gns_set_option(const char* name, int value) {
  if (!name) return GNS_EINVAL;
  GnsTuning& t = tuning();
  if (!std::strcmp(name, "fwd_mapping")) { if (value < 0 || value > 2) return GNS_EINVAL; t.fwd_mapping = value; return GNS_OK; }
  if (!std::strcmp(name, "bwd_variant")) { if (value < 1 || value > 4) return GNS_EINVAL; t.bwd_variant = value; return GNS_OK; }
  if (!std::strcmp(name, "train_mapping")) { if (value < 0 || value > 2) return GNS_EINVAL; t.train_mapping = value; return GNS_OK; }
  if (!std::strcmp(name, "gw_pack")) { if (value < 0 || value > 16) return GNS_EINVAL; t.gw_pack = value; return GNS_OK; }
  if (!std::strcmp(name, "fwd_waves")) { if (value <= 0 || (value & (value - 1)) || value * 64 > GNS_FWD_MAX_THREADS) return GNS_EINVAL; t.fwd_waves = value; return GNS_OK; }
  if (!std::strcmp(name, "fwd_plane")) { if (value < 0 || value > 2) return GNS_EINVAL; t.fwd_plane = value; return GNS_OK; }
  if (!std::strcmp(name, "dw_mfma")) { t.dw_mfma = value ? 1 : 0; return GNS_OK; }
  if (!std::strcmp(name, "team")) { if (value < 0 || value > GNS_MAX_TEAM) return GNS_EINVAL; t.team = value; return GNS_OK; }
  if (!std::strcmp(name, "bwds_mode")) { if (value < 0 || value > 2) return GNS_EINVAL; t.bwds_mode = value; return GNS_OK; }
  if (!std::strcmp(name, "bwds_chunks")) { if (value != 0 && gns_part_index(value) < 0) return GNS_EINVAL; t.bwds_chunks = value; return GNS_OK; }
  return GNS_EINVAL;
}
extern "C" int gns_get_option(const char* name, int* value) {
  if (!name || !value) return GNS_EINVAL;
  const GnsTuning& t = tuning();
  if (!std::strcmp(name, "fwd_mapping")) *value = t.fwd_mapping;
  else if (!std::strcmp(name, "train_mapping")) *value = t.train_mapping;
  else if (!std::strcmp(name, "bwd_variant")) *value = t.bwd_variant;
  else if (!std::strcmp(name, "gw_pack")) *value = t.gw_pack;
  else if (!std::strcmp(name, "fwd_waves")) *value = t.fwd_waves;
  else if (!std::strcmp(name, "fwd_plane")) *value = t.fwd_plane;
  else if (!std::strcmp(name, "dw_mfma")) *value = t.dw_mfma;
  else if (!std::strcmp(name, "team")) *value = t.team;
  else if (!std::strcmp(name, "bwds_chunks")) *value = t.bwds_chunks;
  else if (!std::strcmp(name, "bwds_mode")) *value = t.bwds_mode;
  else return GNS_EINVAL;
  return GNS_OK;
}

// The compiled (latent_dim, hidden_dim) pair a model runs on: the smallest one that holds it.  A narrower model runs zero-padded
// (gns_common.h, GnsFamilies): same function, same gradients; only gns_pack_params / gns_unfold know the difference.
static bool kernel_dims(int d, int h, int* dk, int* hk) {
  bool found = false;
#define GNS_CASE(DD, HH) if (d <= DD && h <= HH && (!found || DD * HH < *dk * *hk)) { *dk = DD; *hk = HH; found = true; }
  GNS_FOR_EACH_DIMS(GNS_CASE)
#undef GNS_CASE
  return found;
}
static bool dims_supported(int d, int h) { int dk, hk; return kernel_dims(d, h, &dk, &hk); }
// cfg with the kernel's dims in place of the model's (everything but the flat parameter layout is sized by these)
static bool kernel_config(const gns_config* model, gns_config* k) {
  *k = *model;
  return kernel_dims(model->latent_dim, model->hidden_dim, &k->latent_dim, &k->hidden_dim);
}

static int check_cfg(const gns_config* c) {
  if (!c) return GNS_EINVAL;
  if (c->n_bus <= 0 || c->n_line <= 0 || c->n_gen < 0 || c->K <= 0 || c->latent_dim <= 0 || c->hidden_dim <= 0) return GNS_EINVAL;
  if (c->multiple_phi != 0 && c->multiple_phi != 1) return GNS_EINVAL;
  return GNS_OK;
}

// ---- optional kernel timing (diagnostics) ---------------------------------------------------------------
#include <vector>
namespace {
struct ProfRing { std::vector<hipEvent_t> a, b; int used = 0; };
ProfRing g_prof[2];
int g_prof_cap = 0;
void prof_mark(int which, bool start, hipStream_t st) {
  if (g_prof_cap <= 0) return;
  ProfRing& r = g_prof[which];
  if (r.used >= g_prof_cap) return;
  (void)hipEventRecord(start ? r.a[r.used] : r.b[r.used], st);
  if (!start) ++r.used;
}
}  // namespace

// Workgroups per 64-grid group of the lane-per-grid kernels (gns_device.h, "teams").  Asked by gns_workspace_bytes,
// gns_forward and gns_backward alike.
static int lane_team(int64_t Bt) {
  const GnsTuning& T = tuning();
  const int64_t groups = (Bt + GNS_LANES - 1) / GNS_LANES;
  if (groups > GNS_TEAM_MAX_GROUPS) return 1;
  const int ncu = device().ncu < GNS_BWD_MAX_WG ? device().ncu : GNS_BWD_MAX_WG;
  return gns_team_size(groups, ncu, T.team);
}

// The split backward (bwd_variant 4) runs the three-phi models on the matrix-pipe engine; everything else keeps the persistent kernel.
static bool use_split_backward(const gns_config* c) {
  const GnsTuning& T = tuning();
  if (!device().split_ready || !gns_bwds_supported(c->latent_dim, c->hidden_dim, c->multiple_phi)) return false;
  if (!gns_backward_persistent_supported(c->latent_dim, c->hidden_dim)) return true;      // the only lane-per-grid backward of this pair
  return T.bwd_variant == 4 && T.dw_mfma;
}

// Which mapping runs a training-mode forward and its backward.  Evaluated identically by gns_forward and gns_backward:
// changing "train_mapping" / "gw_pack" between a forward and its backward is a caller error.
static int gw_train_pack(const gns_config* c, int64_t Bt) {
  const GnsTuning& T = tuning();
  const int P = T.gw_pack > 0 ? T.gw_pack : 1;
  if (!device().gw_ready || T.train_mapping == 1) return 0;
  if (!gns_gw_supported(c->n_bus, c->n_line, c->latent_dim, c->hidden_dim, c->multiple_phi, P)) return 0;
  if (!gns_gw_backward_supported(c->n_bus, c->n_line, c->latent_dim, c->hidden_dim, c->multiple_phi, P)) return 0;
  if (T.train_mapping == 2) return P;
  // auto: below ~2000 grids per GPU (case118; ~1000 for case300) the grid-per-workgroup pair, one workgroup per grid, is the
  // faster one (measured, case118 x 1024: 0.55 vs 1.20 ms); above, the lane-per-grid pair - teams of workgroups per 64-grid
  // group keep the chip busy down to a quarter of its CUs in groups (case118 x 4096: 1.38 vs 1.96 ms, x 8192: 2.31 vs 3.86).
  const int64_t groups = (Bt + GNS_LANES - 1) / GNS_LANES;
  const int wpg = gns_gw_backward_wpg(c->n_bus);
  return groups <= 32 / (wpg > 2 ? wpg / 2 : 1) ? P : 0;
}
// Grids per workgroup of the evaluation-mode forward when it runs on the grid-per-workgroup kernel, 0 when the lane-per-grid
// kernel runs it.  gns_workspace_bytes, gns_forward and gns_uses_packed_inputs must agree, so they all ask here.
static int gw_eval_pack(const gns_config* c, int64_t Bt) {
  const GnsTuning& T = tuning();
  const int N = c->n_bus, E = c->n_line;
  const int wpg = ((N > E ? N : E) + 63) / 64;
  int P = T.gw_pack;
  if (P <= 0) {                                   // auto: one workgroup per CU with as many grids as fit (all waves of a CU in the
    P = 1;                                        // same phase stream the same weights: 73 % scalar-cache hits against 55 %)
    for (int q = 2; q * wpg <= 12; ++q) if (gns_gw_supported(N, E, c->latent_dim, c->hidden_dim, c->multiple_phi, q)) P = q;
  }
  const bool can = device().gw_ready && gns_gw_supported(N, E, c->latent_dim, c->hidden_dim, c->multiple_phi, P);
  bool want = T.fwd_mapping == 2 || (T.fwd_mapping == 0 && N >= 48);
  if (T.fwd_mapping == 0 && want) {
    // auto, by batch: once the lane-per-grid kernel fills the chip with one workgroup per 64-grid group it is the faster one (since it
    // skips the v family on generator buses: case118 x 16384, 256 groups: 0.73 against 0.91 ms); a case too large to pack several
    // grids into a workgroup hands over earlier (case300 x 8192, 128 groups on teams of two: 2.75 against 5.03 ms); small batches
    // stay on chip (case30 x 4096, 64 groups: 0.12 against 0.18 ms).  Measured on MI355X, tools/gpu_time_eval.py.
    const int64_t groups = (Bt + GNS_LANES - 1) / GNS_LANES;
    const int ncu = device().ncu > 0 ? device().ncu : 256;
    if (groups >= ncu || (P == 1 && wpg >= 5 && groups * 4 >= ncu)) want = false;
  }
  return (can && want) ? P : 0;
}

struct GwTrainLayout { size_t off_pt, off_pn, off_save; GwSaveLayout sv; size_t fwd_total; int blocks, waves; size_t off_slab, off_part, off_tmp, bwd_total; long long slab_floats, nslab; };
static GwTrainLayout gw_train_layout(const gns_config* c, int64_t Bt, int P) {
  GwTrainLayout L;
  GnsFamilies f; gns_families(c->latent_dim, c->hidden_dim, c->K, c->multiple_phi, &f);
  size_t o = 0;
  L.off_pt = o; o = gns_align256(o + (size_t)f.t_total * 4);
  L.off_pn = o; o = gns_align256(o + (size_t)f.n_total * 4);
  L.off_save = o;
  L.sv = gw_save_layout(c->n_bus, c->latent_dim, c->hidden_dim, c->K, c->multiple_phi, Bt);
  L.fwd_total = o + L.sv.total;
  const int WPG = gns_gw_backward_wpg(c->n_bus);
  L.blocks = gns_gw_backward_blocks(c->n_bus, c->n_line, c->latent_dim, c->hidden_dim, c->multiple_phi, P, Bt);
  L.waves = P * WPG;
  L.slab_floats = (f.g_total + 63) / 64 * 64;
  L.nslab = (long long)L.blocks * L.waves;
  o = 0;
  L.off_slab = o; o = gns_align256(o + (size_t)L.nslab * L.slab_floats * 4);
  L.off_part = o; o = gns_align256(o + (size_t)GNS_RED_PARTS * L.slab_floats * 4);
  L.off_tmp = o;  o = gns_align256(o + (size_t)L.slab_floats * 4);
  L.bwd_total = o;
  return L;
}

extern "C" int gns_profile_enable(int capacity) {
  for (auto& r : g_prof) {
    for (auto e : r.a) (void)hipEventDestroy(e);
    for (auto e : r.b) (void)hipEventDestroy(e);
    r.a.clear(); r.b.clear(); r.used = 0;
  }
  g_prof_cap = 0;
  if (capacity < 0) return GNS_EINVAL;
  for (auto& r : g_prof)
    for (int i = 0; i < capacity; ++i) {
      hipEvent_t x, y;
      if (hipEventCreate(&x) != hipSuccess || hipEventCreate(&y) != hipSuccess) return GNS_ELAUNCH;
      r.a.push_back(x); r.b.push_back(y);
    }
  g_prof_cap = capacity;
  return GNS_OK;
}

extern "C" int gns_profile_read(int backward, float* ms_sum, int* launches) {
  if (!ms_sum || !launches) return GNS_EINVAL;
  ProfRing& r = g_prof[backward ? 1 : 0];
  float tot = 0.f;
  for (int i = 0; i < r.used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b[i]) != hipSuccess || hipEventElapsedTime(&ms, r.a[i], r.b[i]) != hipSuccess) return GNS_ELAUNCH;
    tot += ms;
  }
  *ms_sum = tot; *launches = r.used; r.used = 0;
  return GNS_OK;
}

extern "C" const char* gns_version(void) { return "gns_hip 0.1 gfx950"; }

extern "C" int gns_param_count(const gns_config* cfg, int64_t* count) {
  if (!cfg || !count || cfg->K <= 0 || cfg->latent_dim <= 0 || cfg->hidden_dim <= 0) return GNS_EINVAL;
  GnsFamilies f; gns_families(cfg->latent_dim, cfg->hidden_dim, cfg->K, cfg->multiple_phi, &f);
  *count = f.flat_total;
  return GNS_OK;
}

extern "C" int gns_config_supported(const gns_config* cfg) {
  if (check_cfg(cfg) != GNS_OK) return 0;
  return dims_supported(cfg->latent_dim, cfg->hidden_dim) && cfg->K <= GNS_MAX_K ? 1 : 0;
}

extern "C" int gns_workspace_bytes(const gns_config* cfg, int64_t Bt, int save_state, size_t* fwd_bytes, size_t* bwd_bytes) {
  int rc = check_cfg(cfg);
  if (rc != GNS_OK) return rc;
  if (Bt <= 0) return GNS_EINVAL;
  const gns_config* model = cfg; gns_config kcfg_;
  if (!kernel_config(model, &kcfg_)) return GNS_EUNSUPPORTED;
  cfg = &kcfg_; (void)model;
  GnsFwdLayout L;
  gns_fwd_layout(cfg->n_bus, cfg->n_line, cfg->latent_dim, cfg->hidden_dim, cfg->K, cfg->multiple_phi, Bt, save_state, &L);
  const int P = save_state ? gw_train_pack(cfg, Bt) : 0;
  if (P > 0) {
    const GwTrainLayout G = gw_train_layout(cfg, Bt, P);
    if (fwd_bytes) *fwd_bytes = G.fwd_total;
    if (bwd_bytes) *bwd_bytes = G.bwd_total;
    return GNS_OK;
  }
  if (!save_state && gw_eval_pack(cfg, Bt) > 0) {                      // state on chip, inputs read in place: only the parameter streams
    if (fwd_bytes) *fwd_bytes = L.off_in;
    if (bwd_bytes) *bwd_bytes = 0;
    return GNS_OK;
  }
  if (fwd_bytes) *fwd_bytes = L.total;
  if (bwd_bytes) {
    GnsBwdLayout B;
    gns_bwd_layout(cfg->n_bus, cfg->n_line, cfg->latent_dim, cfg->hidden_dim, cfg->K, cfg->multiple_phi, Bt, lane_team(Bt), &B);
    *bwd_bytes = B.total;
    if (device().split_ready && gns_bwds_supported(cfg->latent_dim, cfg->hidden_dim, cfg->multiple_phi)) {   // either variant may be asked for later
      GnsBwdsLayout S;
      gns_bwds_layout(cfg->n_bus, cfg->n_line, cfg->latent_dim, cfg->hidden_dim, cfg->K, cfg->multiple_phi, Bt, device().ncu, tuning().bwds_chunks, &S);
      if (S.total > *bwd_bytes) *bwd_bytes = S.total;
    }
  }
  return GNS_OK;
}

extern "C" int gns_uses_packed_inputs(const gns_config* cfg, int64_t Bt, int save_state) {
  if (check_cfg(cfg) != GNS_OK || Bt <= 0) return 0;
  const gns_config* model = cfg; gns_config kcfg_;
  if (!kernel_config(model, &kcfg_)) return 0;
  cfg = &kcfg_; (void)model;
  return (save_state ? gw_train_pack(cfg, Bt) : gw_eval_pack(cfg, Bt)) > 0 ? 0 : 1;
}

// Did a team of workgroups give up at a barrier during the gns_forward that used this workspace?  (Teams: lane-per-grid kernels on a
// batch with fewer 64-grid groups than CUs; a barrier gives up after ~seconds when a partner workgroup never became resident -
// another kernel or process holding its CU.  The losses of that call are NaN; this is how the host learns it without looking at
// them.)  The ONE entry point that synchronises: it waits for `stream` and copies one word.  *status = 0 without touching the
// device when this (cfg, Bt) does not use teams.
// Byte offset of that status word inside the forward workspace, or (size_t)-1 in *offset when this (cfg, Bt, save_state) runs
// without teams.  (Tests inject a failure through it; the host wrapper asks it whether a status has to be checked at all.)
extern "C" int gns_team_status_offset(const gns_config* cfg, int64_t Bt, int save_state, size_t* offset) {
  int rc = check_cfg(cfg);
  if (rc != GNS_OK) return rc;
  if (!offset || Bt <= 0) return GNS_EINVAL;
  const gns_config* model = cfg; gns_config kcfg_;
  if (!kernel_config(model, &kcfg_)) return GNS_EUNSUPPORTED;
  cfg = &kcfg_; (void)model;
  *offset = (size_t)-1;
  if (lane_team(Bt) <= 1) return GNS_OK;
  if ((save_state ? gw_train_pack(cfg, Bt) : gw_eval_pack(cfg, Bt)) > 0) return GNS_OK;          // the grid-per-workgroup kernels have no teams
  GnsFwdLayout L;
  gns_fwd_layout(cfg->n_bus, cfg->n_line, cfg->latent_dim, cfg->hidden_dim, cfg->K, cfg->multiple_phi, Bt, save_state, &L);
  *offset = L.off_team + GNS_TEAM_STATUS_WORD * 4;
  return GNS_OK;
}

extern "C" int gns_team_status(const gns_config* cfg, int64_t Bt, const void* fwd_workspace, size_t fwd_workspace_bytes, int save_state,
                               int* status, void* stream) {
  int rc = check_cfg(cfg);
  if (rc != GNS_OK) return rc;
  if (!status || !fwd_workspace || Bt <= 0) return GNS_EINVAL;
  const gns_config* model = cfg; gns_config kcfg_;
  if (!kernel_config(model, &kcfg_)) return GNS_EUNSUPPORTED;
  cfg = &kcfg_; (void)model;
  *status = 0;
  if (lane_team(Bt) <= 1) return GNS_OK;
  if ((save_state ? gw_train_pack(cfg, Bt) : gw_eval_pack(cfg, Bt)) > 0) return GNS_OK;          // the grid-per-workgroup kernels have no teams
  GnsFwdLayout L;
  gns_fwd_layout(cfg->n_bus, cfg->n_line, cfg->latent_dim, cfg->hidden_dim, cfg->K, cfg->multiple_phi, Bt, save_state, &L);
  if (fwd_workspace_bytes < L.total) return GNS_ESIZE;
  unsigned word = 0;
  if (hipMemcpyAsync(&word, (const char*)fwd_workspace + L.off_team + GNS_TEAM_STATUS_WORD * 4, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { (void)hipGetLastError(); return GNS_ELAUNCH; }
  *status = word ? 1 : 0;
  return GNS_OK;
}

extern "C" int gns_prepack_bytes(const gns_config* cfg, int64_t Bt, size_t* bytes) {
  int rc = check_cfg(cfg);
  if (rc != GNS_OK) return rc;
  if (!bytes || Bt <= 0) return GNS_EINVAL;
  const int64_t groups = (Bt + GNS_LANES - 1) / GNS_LANES;
  *bytes = (size_t)groups * gns_in_rows(cfg->n_bus, cfg->n_line) * GNS_LANES * 16;
  return GNS_OK;
}

extern "C" int gns_prepack(const gns_config* cfg, const void* topo_dev, const float* buses, const float* lines,
                           const float* generators, int64_t Bt, void* packed, size_t packed_bytes, void* stream) {
  size_t need = 0;
  int rc = gns_prepack_bytes(cfg, Bt, &need);
  if (rc != GNS_OK) return rc;
  if (!topo_dev || !buses || !lines || !generators || !packed) return GNS_EINVAL;
  if (packed_bytes < need) return GNS_ESIZE;
  return gns_launch_pack_inputs((const int*)topo_dev, buses, lines, generators, (float*)packed, cfg->n_bus, cfg->n_line, cfg->n_gen, Bt,
                                (Bt + GNS_LANES - 1) / GNS_LANES, (hipStream_t)stream);
}

extern "C" int gns_forward(const gns_config* cfg, const void* topo_dev, const float* params, const float* buses,
                           const float* lines, const float* generators, int64_t Bt, const void* packed_inputs, float* v, float* theta,
                           float* total_loss, float* last_loss, void* workspace, size_t workspace_bytes, int save_state,
                           void* stream) {
  int rc = check_cfg(cfg);
  if (rc != GNS_OK) return rc;
  if (!topo_dev || !params || !buses || !lines || !generators || !v || !theta || !total_loss || !last_loss || !workspace || Bt <= 0)
    return GNS_EINVAL;
  const gns_config* model = cfg; gns_config kcfg_;
  if (!kernel_config(model, &kcfg_) || cfg->K > GNS_MAX_K) return GNS_EUNSUPPORTED;
  cfg = &kcfg_;                          // the kernel's dims from here on; the model's only lay out the flat parameters
  const int N = cfg->n_bus, E = cfg->n_line, Gn = cfg->n_gen, K = cfg->K, d = cfg->latent_dim, h = cfg->hidden_dim;
  GnsFamilies fam; gns_families_padded(model->latent_dim, model->hidden_dim, d, h, K, cfg->multiple_phi, &fam);
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const GnsTuning& T = tuning();
  if (const int TP = save_state ? gw_train_pack(cfg, Bt) : 0) {          // training-mode forward of the grid-per-workgroup pair
    const GwTrainLayout GL = gw_train_layout(cfg, Bt, TP);
    if (workspace_bytes < GL.fwd_total) return GNS_ESIZE;
    float* gpt = (float*)(ws + GL.off_pt);
    float* gpn = (float*)(ws + GL.off_pn);
    rc = gns_launch_pack_params(params, gpt, gpn, fam, K, d, h, st);
    if (rc != GNS_OK) return rc;
    GnsGwFwdArgs G;
    std::memset(&G, 0, sizeof(G));
    G.topo = (const int*)topo_dev; G.pt = gpt; G.buses = buses; G.lines = lines; G.gens = generators;
    G.v_out = v; G.theta_out = theta; G.total_out = total_loss; G.last_out = last_loss;
    char* sv = ws + GL.off_save;
    G.sv_state = (float*)(sv + GL.sv.off_state); G.sv_S = (float*)(sv + GL.sv.off_S); G.sv_lam = (float*)(sv + GL.sv.off_lam);
    for (int i = 0; i < fam.nfam; ++i) { G.t_off[i] = fam.t_off[i]; G.t_sz[i] = fam.t_sz[i]; }
    for (int k = 0; k < K; ++k) G.gw[k] = (float)std::pow((double)cfg->gamma, (double)(K - k));
    G.Bt = Bt; G.N = N; G.E = E; G.Gn = Gn; G.K = K; G.save = 1; G.P = TP; G.WPG = ((N > E ? N : E) + 63) / 64;
    prof_mark(0, true, st);
    rc = gns_gw_launch_forward(d, h, cfg->multiple_phi, G, st);
    prof_mark(0, false, st);
    return rc;
  }
  GnsFwdLayout L;
  gns_fwd_layout(N, E, d, h, K, cfg->multiple_phi, Bt, save_state, &L);
  if (workspace_bytes < ((!save_state && gw_eval_pack(cfg, Bt) > 0) ? L.off_in : L.total)) return GNS_ESIZE;
  float* pt = (float*)(ws + L.off_pt);
  float* pn = (float*)(ws + L.off_pn);
  float* pin = (float*)(ws + L.off_in);
  rc = gns_launch_pack_params(params, pt, pn, fam, K, d, h, st);
  if (rc != GNS_OK) return rc;
  // Evaluation (nothing saved for a backward): the grid-per-workgroup mapping keeps the whole state on chip.
  {
    const int P = save_state ? 0 : gw_eval_pack(cfg, Bt);
    if (P > 0) {
      GnsGwFwdArgs G;
      std::memset(&G, 0, sizeof(G));
      G.topo = (const int*)topo_dev; G.pt = pt; G.buses = buses; G.lines = lines; G.gens = generators;
      G.v_out = v; G.theta_out = theta; G.total_out = total_loss; G.last_out = last_loss;
      for (int i = 0; i < fam.nfam; ++i) { G.t_off[i] = fam.t_off[i]; G.t_sz[i] = fam.t_sz[i]; }
      for (int k = 0; k < K; ++k) G.gw[k] = (float)std::pow((double)cfg->gamma, (double)(K - k));   // main.py:198
      G.Bt = Bt; G.N = N; G.E = E; G.Gn = Gn; G.K = K; G.save = 0; G.P = P; G.WPG = ((N > E ? N : E) + 63) / 64;
      prof_mark(0, true, st);
      rc = gns_gw_launch_forward(d, h, cfg->multiple_phi, G, st);
      prof_mark(0, false, st);
      return rc;
    }
  }
  if (packed_inputs) pin = (float*)packed_inputs;                  // a resident batch packed once by gns_prepack: nothing to redo
  else {
    rc = gns_launch_pack_inputs((const int*)topo_dev, buses, lines, generators, pin, N, E, Gn, Bt, L.groups, st);
    if (rc != GNS_OK) return rc;
  }
  GnsFwdArgs A;
  std::memset(&A, 0, sizeof(A));
  A.topo = (const int*)topo_dev; A.pt = pt; A.in = pin;
  A.state = (float*)(ws + L.off_state); A.lam = (float*)(ws + L.off_lam); A.msg = (float*)(ws + L.off_msg);
  A.v_out = v; A.theta_out = theta; A.total_out = total_loss; A.last_out = last_loss;
  for (int i = 0; i < fam.nfam; ++i) { A.t_off[i] = fam.t_off[i]; A.t_sz[i] = fam.t_sz[i]; }
  for (int k = 0; k < K; ++k) A.gw[k] = (float)std::pow((double)cfg->gamma, (double)(K - k));   // main.py:198
  A.Bt = Bt; A.G = L.groups; A.N = N; A.E = E; A.K = K; A.save = save_state ? 1 : 0;
  const int team0 = lane_team(Bt);
  A.team = team0;
  A.team_ws = (unsigned char*)(ws + L.off_team);
  int waves = T.fwd_waves;
  while (waves * A.team > GNS_MAXP) waves /= 2;
  A.part_idx = gns_part_index(waves * A.team);
  auto pick_planes = [&]() {
    A.plane = (device().fwd_ready && gns_fwd_plane_fits(N, A.team) && T.fwd_plane) ? 1 : 0;
    if (A.plane && gns_fwd_plane2_fits(N, A.team) && T.fwd_plane == 2) A.plane = 2;
  };
  pick_planes();
  if (A.team > 1) {
    // every workgroup of every team must be resident at once: the kernel's own occupancy at this launch configuration says
    // how many a CU holds (not just the CU count); a configuration that does not fit runs one workgroup per group instead
    const int per_cu = gns_fwd_blocks_per_cu(d, h, cfg->multiple_phi, A, waves * 64);
    if ((long long)per_cu * device().ncu < A.G * A.team) {
      A.team = 1;
      waves = T.fwd_waves;
      A.part_idx = gns_part_index(waves);
      pick_planes();
    }
  }
  // (the counters - and the status word gns_team_status reads - are zeroed whenever this batch size is one that may use teams)
  if (team0 > 1 && hipMemsetAsync(A.team_ws, 0, (size_t)L.groups * GNS_TEAM_CTR_BYTES, st) != hipSuccess) return GNS_ELAUNCH;
  prof_mark(0, true, st);
  rc = gns_launch_forward(d, h, cfg->multiple_phi, A, waves * 64, st);
  prof_mark(0, false, st);
  return rc;
}

extern "C" int gns_backward(const gns_config* cfg, const void* topo_dev, const float* params,
                            const float* buses, const float* lines, const float* generators, int64_t Bt, const void* packed_inputs,
                            const void* fwd_workspace, size_t fwd_workspace_bytes, const float* grad_total,
                            const float* grad_last, const float* grad_v, const float* grad_theta, float* grad_params,
                            void* bwd_workspace, size_t bwd_workspace_bytes, void* stream) {
  int rc = check_cfg(cfg);
  if (rc != GNS_OK) return rc;
  if (!topo_dev || !params || !fwd_workspace || !grad_params || !bwd_workspace || Bt <= 0) return GNS_EINVAL;
  const gns_config* model = cfg; gns_config kcfg_;
  if (!kernel_config(model, &kcfg_) || cfg->K > GNS_MAX_K) return GNS_EUNSUPPORTED;
  cfg = &kcfg_;                          // the kernel's dims from here on; the model's only lay out the flat parameters and their gradient
  const int N = cfg->n_bus, E = cfg->n_line, K = cfg->K, d = cfg->latent_dim, h = cfg->hidden_dim;
  if (const int TP = gw_train_pack(cfg, Bt)) {                            // the pair of the grid-per-workgroup training forward
    if (!buses || !lines || !generators) return GNS_EINVAL;
    const GwTrainLayout GL = gw_train_layout(cfg, Bt, TP);
    if (fwd_workspace_bytes < GL.fwd_total || bwd_workspace_bytes < GL.bwd_total) return GNS_ESIZE;
    GnsFamilies fam; gns_families_padded(model->latent_dim, model->hidden_dim, d, h, K, cfg->multiple_phi, &fam);
    hipStream_t st = (hipStream_t)stream;
    const char* fw = (const char*)fwd_workspace;
    char* bw = (char*)bwd_workspace;
    if (hipMemsetAsync(bw + GL.off_slab, 0, (size_t)GL.nslab * GL.slab_floats * 4, st) != hipSuccess) return GNS_ELAUNCH;
    GnsGwBwdArgs G;
    std::memset(&G, 0, sizeof(G));
    G.topo = (const int*)topo_dev;
    G.pt = (const float*)(fw + GL.off_pt); G.pn = (const float*)(fw + GL.off_pn);
    G.buses = buses; G.lines = lines; G.gens = generators;
    const char* sv = fw + GL.off_save;
    G.sv_state = (const float*)(sv + GL.sv.off_state); G.sv_S = (const float*)(sv + GL.sv.off_S); G.sv_lam = (const float*)(sv + GL.sv.off_lam);
    G.g_total = grad_total; G.g_last = grad_last; G.g_v = grad_v; G.g_theta = grad_theta;
    G.slab = (float*)(bw + GL.off_slab);
    for (int i = 0; i < fam.nfam; ++i) {
      G.t_off[i] = fam.t_off[i]; G.t_sz[i] = fam.t_sz[i]; G.n_off[i] = fam.n_off[i]; G.n_sz[i] = fam.n_sz[i];
      G.g_off[i] = fam.g_off[i]; G.g_sz[i] = fam.g_sz[i];
    }
    for (int k = 0; k < K; ++k) G.gw[k] = (float)std::pow((double)cfg->gamma, (double)(K - k));
    G.Bt = Bt; G.slab_floats = GL.slab_floats; G.N = N; G.E = E; G.Gn = cfg->n_gen; G.K = K;
    G.P = TP; G.WPG = gns_gw_backward_wpg(N);
    prof_mark(1, true, st);
    rc = gns_gw_launch_backward(d, h, cfg->multiple_phi, G, GL.blocks, st);
    prof_mark(1, false, st);
    if (rc != GNS_OK) return rc;
    return gns_launch_reduce(G.slab, (float*)(bw + GL.off_part), (float*)(bw + GL.off_tmp), params, grad_params, GL.nslab,
                             GL.slab_floats, fam, K, d, h, st);
  }
  GnsFwdLayout L;
  gns_fwd_layout(N, E, d, h, K, cfg->multiple_phi, Bt, 1, &L);
  if (use_split_backward(cfg)) {
    GnsBwdsLayout S;
    gns_bwds_layout(N, E, d, h, K, cfg->multiple_phi, Bt, device().ncu, tuning().bwds_chunks, &S);
    if (fwd_workspace_bytes < L.total || bwd_workspace_bytes < S.total) return GNS_ESIZE;
    GnsFamilies fam; gns_families_padded(model->latent_dim, model->hidden_dim, d, h, K, cfg->multiple_phi, &fam);
    hipStream_t st = (hipStream_t)stream;
    const char* fw = (const char*)fwd_workspace;
    char* bw = (char*)bwd_workspace;
    GnsBwdsArgs A;
    std::memset(&A, 0, sizeof(A));
    A.topo = (const int*)topo_dev;
    A.pt = (const float*)(fw + L.off_pt); A.pn = (const float*)(fw + L.off_pn);
    A.in = packed_inputs ? (const float*)packed_inputs : (const float*)(fw + L.off_in);
    A.state = (const float*)(fw + L.off_state); A.lam = (const float*)(fw + L.off_lam); A.msg = (const float*)(fw + L.off_msg);
    A.g_total = grad_total; A.g_last = grad_last; A.g_v = grad_v; A.g_theta = grad_theta;
    A.adj = (float*)(bw + S.off_adj); A.slots = (float*)(bw + S.off_slots); A.slab = (float*)(bw + S.off_slab);
    for (int i = 0; i < fam.nfam; ++i) {
      A.t_off[i] = fam.t_off[i]; A.t_sz[i] = fam.t_sz[i]; A.n_off[i] = fam.n_off[i]; A.n_sz[i] = fam.n_sz[i];
      A.g_off[i] = fam.g_off[i]; A.g_sz[i] = fam.g_sz[i];
    }
    A.Bt = Bt; A.G = S.groups; A.slab_floats = S.slab_floats; A.N = N; A.E = E; A.K = K;
    A.C = S.C; A.part_idx = gns_part_index(S.C); A.R = S.R;
    A.RB = (int)(1 + S.mq); A.RBA = (int)S.adj_rows;
    A.mode = cfg->multiple_phi ? tuning().bwds_mode : 2;          // the single phi is reversed after all three L nets: bus-major
    const size_t lds = gns_bwds_phys_lds(N, &A.use_plane);
    prof_mark(1, true, st);
    for (int k = K - 1; k >= 0; --k) {
      A.k = k;
      A.gwk = (float)std::pow((double)cfg->gamma, (double)(K - k));
      rc = gns_launch_bwds_phys(A, lds, st);
      if (rc != GNS_OK) return rc;
      rc = gns_launch_bwds_sweep(d, h, cfg->multiple_phi, A, st);
      if (rc != GNS_OK) return rc;
    }
    prof_mark(1, false, st);
    return gns_launch_reduce(A.slab, (float*)(bw + S.off_part), (float*)(bw + S.off_tmp), params, grad_params, S.nslab, S.slab_floats,
                             fam, K, d, h, st);
  }
  GnsBwdLayout B;
  const int team = lane_team(Bt);
  gns_bwd_layout(N, E, d, h, K, cfg->multiple_phi, Bt, team, &B);
  if (fwd_workspace_bytes < L.total || bwd_workspace_bytes < B.total) return GNS_ESIZE;
  GnsFamilies fam; gns_families_padded(model->latent_dim, model->hidden_dim, d, h, K, cfg->multiple_phi, &fam);
  hipStream_t st = (hipStream_t)stream;
  const char* fw = (const char*)fwd_workspace;
  char* bw = (char*)bwd_workspace;
  const int blocks = (int)(B.groups * team < GNS_BWD_MAX_WG ? B.groups * team : GNS_BWD_MAX_WG);
  const long long nslab = (long long)blocks * GNS_BWD_WAVES;
  // the V2 sweep writes every slab entry itself on a workgroup's first group: no 90 MB memset in front of it
  const bool v2 = tuning().dw_mfma && tuning().bwd_variant >= 2 && cfg->multiple_phi;
  if (!v2 && hipMemsetAsync(bw + B.off_slab, 0, (size_t)nslab * B.slab_floats * 4, st) != hipSuccess) return GNS_ELAUNCH;
  GnsBwdArgs A;
  std::memset(&A, 0, sizeof(A));
  A.topo = (const int*)topo_dev;
  A.pt = (const float*)(fw + L.off_pt); A.pn = (const float*)(fw + L.off_pn);
  A.in = packed_inputs ? (const float*)packed_inputs : (const float*)(fw + L.off_in);
  A.state = (const float*)(fw + L.off_state); A.lam = (const float*)(fw + L.off_lam); A.msg = (const float*)(fw + L.off_msg);
  A.g_total = grad_total; A.g_last = grad_last; A.g_v = grad_v; A.g_theta = grad_theta;
  A.adj = (float*)(bw + B.off_adj); A.slots = (float*)(bw + B.off_slots); A.slab = (float*)(bw + B.off_slab);
  for (int i = 0; i < fam.nfam; ++i) {
    A.t_off[i] = fam.t_off[i]; A.t_sz[i] = fam.t_sz[i]; A.n_off[i] = fam.n_off[i]; A.n_sz[i] = fam.n_sz[i];
    A.g_off[i] = fam.g_off[i]; A.g_sz[i] = fam.g_sz[i];
  }
  for (int k = 0; k < K; ++k) A.gw[k] = (float)std::pow((double)cfg->gamma, (double)(K - k));
  A.Bt = Bt; A.G = B.groups; A.slab_floats = B.slab_floats; A.N = N; A.E = E; A.K = K;
  A.part_idx = gns_part_index(GNS_BWD_WAVES * team);
  A.team = team; A.team_ws = (unsigned char*)(bw + B.off_team);
  if (team > 1 && hipMemsetAsync(A.team_ws, 0, (size_t)B.groups * GNS_TEAM_CTR_BYTES, st) != hipSuccess) return GNS_ELAUNCH;
  A.slab_dirty = v2 ? 1 : 0;
  prof_mark(1, true, st);
  // The weight-gradient contraction over the grids runs on the matrix pipe (exact fp32) unless GNS_DW_MFMA=0 asks for
  // the packed-FMA register tiles; both are parity-tested (gns_backward.hip, "weight-gradient engines").
  rc = gns_launch_backward(d, h, cfg->multiple_phi, tuning().dw_mfma, tuning().bwd_variant, A, blocks, st);
  prof_mark(1, false, st);
  if (rc != GNS_OK) return rc;
  // the kernel has summed each workgroup's eight slabs into its first one: one slab per workgroup is left to reduce
  return gns_launch_reduce(A.slab, (float*)(bw + B.off_part), (float*)(bw + B.off_tmp), params, grad_params, blocks, B.slab_floats,
                           fam, K, d, h, st, (long long)GNS_BWD_WAVES * B.slab_floats);
}

// ---- Adam on the flat parameter buffer (GNS/main.py:290 with the optimiser of main.py:241-243) ---------------------------
__global__ void gns_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                long long n, float one_minus_b1, float b2, float one_minus_b2, float step_size, float inv_sqrt_bc2, float eps) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = m[i] + (gi - m[i]) * one_minus_b1;
  const float vi = b2 * v[i] + one_minus_b2 * gi * gi;
  m[i] = mi; v[i] = vi;
  p[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
}

extern "C" int gns_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             double lr, double beta1, double beta2, double eps, int64_t step, void* stream) {
  if (!params || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return GNS_EINVAL;
  const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
  hipLaunchKernelGGL(gns_adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grad, exp_avg,
                     exp_avg_sq, (long long)n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)(lr / bc1),
                     (float)(1.0 / std::sqrt(bc2)), (float)eps);
  return hipGetLastError() == hipSuccess ? GNS_OK : GNS_ELAUNCH;
}

// The same update with the step counter ON THE DEVICE, so that a whole training step (forward, backward, this) can be captured
// into a HIP graph once and replayed: a captured launch bakes its scalar arguments in, and Adam's bias corrections change every
// step.  state[0] = steps taken so far (a float: exact up to 2^24), state[1..2] = scratch (the step size and 1 / sqrt(1 - beta2^t)
// of the current step).  Two launches: one thread advances the counter and forms the corrections in double like the host
// version, then the element-wise kernel reads them.
__global__ void gns_adam_advance_kernel(float* __restrict__ st, double lr, double b1, double b2) {
  const double step = (double)st[0] + 1.0;
  st[0] = (float)step;
  st[1] = (float)(lr / (1.0 - pow(b1, step)));
  st[2] = (float)(1.0 / sqrt(1.0 - pow(b2, step)));
}
__global__ void gns_adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                    long long n, float one_minus_b1, float b2, float one_minus_b2, const float* __restrict__ st, float eps) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float step_size = st[1], inv_sqrt_bc2 = st[2];
  const float gi = g[i];
  const float mi = m[i] + (gi - m[i]) * one_minus_b1;
  const float vi = b2 * v[i] + one_minus_b2 * gi * gi;
  m[i] = mi; v[i] = vi;
  p[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
}

extern "C" int gns_adam_step_dev(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 double lr, double beta1, double beta2, double eps, float* step_state, void* stream) {
  if (!params || !grad || !exp_avg || !exp_avg_sq || !step_state || n <= 0) return GNS_EINVAL;
  hipLaunchKernelGGL(gns_adam_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_state, lr, beta1, beta2);
  hipLaunchKernelGGL(gns_adam_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grad, exp_avg,
                     exp_avg_sq, (long long)n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_state, (float)eps);
  return hipGetLastError() == hipSuccess ? GNS_OK : GNS_ELAUNCH;
}
