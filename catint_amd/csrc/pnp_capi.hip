// C-ABI of the batched PNP transport path (include/catint_pnp.h): handle, device buffers, the
// integrate_pnp loop (reference catint/calculator_old.py:210, :512, :990) around the HIP kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pnp_internal.h"
#include "pnp_step_table.h"

using namespace pnp;

struct pnp_handle {
  pnp_config cfg;
  DevArgs a;
  ReactionTable rt;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int P = 0;
  int64_t B = 0;
  bool have_species = false, have_batch = false;
  int64_t steps_done = 0;
  // device buffers (capacity-sized)
  double *c = nullptr, *lapl[2] = {nullptr, nullptr}, *v = nullptr, *gradv = nullptr, *rates = nullptr;
  double *pb = nullptr, *vzeta = nullptr, *flux = nullptr, *cbulk = nullptr, *csurf = nullptr;
  double *ytmp = nullptr, *ftmp = nullptr;   // method-of-lines scratch: state in, derivative out
  double* mol_lapl = nullptr;                // ... and its charge row for grids beyond one wave
  double* ode_buf = nullptr;                 // pnp_integrate_dopri5 / _dop853: k buffers, y1, ysti ([cap][N][ldx] each) + per-lane reals
  int ode_nbuf = 0;                          // state-sized buffers in ode_buf (8 DOPRI5, 12 DOP853)
  int32_t* ode_int = nullptr;                // ... per-lane integers + 64 counters
  double* rkc_d = nullptr;                   // pnp_integrate_rkc: per-lane reals [cap][RKC_ND] (the state-sized buffers are ode_buf's)
  int32_t* rkc_i = nullptr;                  // ... per-lane integers [cap][RKC_NI] + 64 counters
  double* stage = nullptr;                   // upload staging [B][N][nx] (pnp_set_batch)
  SpecConst* spec = nullptr;
  Options opt;               // debug / tuning switches: pnp_set_option; defaults from the CATINT_* environment at pnp_create
  int32_t* status = nullptr;
  // physical mode (PNP_METHOD_NEWTON)
  bool newton = false;
  pnp_newton_params np;
  double Dk[PNP_NEWTON_MAX_SPECIES] = {0}, qk[PNP_NEWTON_MAX_SPECIES] = {0}, volk[PNP_NEWTON_MAX_SPECIES] = {0};
  bool mpb = false;
  double velocity = 0.0;                 // pnp_set_convection
  double* c_old = nullptr;
  double* work = nullptr;
  double* stash = nullptr;
  ReactionTable* rt_dev = nullptr;
  double* c_old2 = nullptr;                 // BDF2 (pnp_newton_params.time_order = 2): the time level before the previous one
  int32_t* bdf_acc = nullptr;               // ... and per-lane iteration counts / status summed over the launches of one pnp_step call
  double* phi_old2 = nullptr;               // predictor: the potential of the time level before the current one
  double* vol_dev = nullptr;                // ... and the ion volumes for its crowding guard
  int nw_ext_old = 0;                       // set around a prepared step: c_old is given (BDF2 combination or c_n under the predictor)
  double nw_sig_scale = 1.0;                // ... and 1/dt carries the factor 3/2 of BDF2
  int nw_bdf2_inline = 0, nw_bdf_hist0 = 0; // set around a launch of several BDF2 steps by a lane kernel, which keeps the history itself
  bool bdf_history = false;                 // c_old2 holds the level before the current state (false after an upload, a change of
                                            // time_order, a stationary solve or patched lanes: the next step is backward Euler)
  ReactionSides* rs_dev = nullptr;          // the table flattened per reaction side (lane kernels)
  int rs_max_exponent = 0;
  int n_wk = 0;
  int32_t wk_species[PNP_MAX_WALL_REACTIONS] = {0};
  double wk_nu[PNP_MAX_WALL_REACTIONS][PNP_NEWTON_MAX_SPECIES] = {{0}};
  double* wk_k = nullptr;
  double wk_alpha[PNP_MAX_WALL_REACTIONS] = {0}, wk_sat[PNP_MAX_WALL_REACTIONS] = {0};     // pnp_set_wall_rate_law
  double *gw = nullptr, *gv = nullptr;
  std::vector<double> xgrid;
  int64_t stash_stride = 0;
  int64_t work_stride = 0;
  int32_t* iters = nullptr;
  int nw_blocks = 0;
  // device SCF loop (pnp_scf_cycle): per-lane bookkeeping, allocated on first use; the two switches below are set around
  // its transport solves
  double* sweep = nullptr;               // sweep kernel: records of the resident teams (allocated on first use)
  int sweep_blocks = 0;
  double* lane_buf = nullptr;            // lane kernel: batch-innermost state copies + records of lane_groups groups of 32 operating points
  int64_t lane_groups = 0;
  double* lane2_buf = nullptr;           // lane-pair kernel: the same for groups of 16
  int64_t lane2_groups = 0;
  double* lane4_buf = nullptr;           // lane-quad kernel: the same for groups of 8
  int64_t lane4_groups = 0;
  // lane kernels: which operating point a slot (group, lane) holds -- points ordered by expected Newton iterations, see lane_order
  int32_t* lane_perm = nullptr;          // device [capacity]
  std::vector<float> lane_key;           // |phiM - phi_bulk| per operating point (pnp_set_batch / pnp_set_pb): the order of a first call
  std::vector<int32_t> lane_iters_host, lane_perm_host;
  bool iters_valid = false;              // h->iters holds the iteration counts of a Newton call on this batch
  bool lane_perm_keep = false;           // inside one pnp_step call of several launches (BDF2 / predictor): the launches after the first
  int64_t lane_perm_B = 0;               // keep the first one's order (no read-back, no sort, no synchronisation per timestep)
  double* scf_d = nullptr;               // (4N + 5) B doubles
  double* scf_snap = nullptr;            // (N + 1) ldx B doubles: per-lane state of the last converged transport solve
  int32_t* scf_i = nullptr;              // 3 B flags + 65 counters
  const int32_t* newton_mask = nullptr;  // lanes to solve (null: all)
  int32_t* user_mask = nullptr;          // pnp_set_lane_mask's copy
  std::vector<int32_t> user_mask_host;   // ... and on the host: a masked solve is sized (kernel choice, lane groups) by the lanes it solves
  int64_t user_mask_count = 0;
  bool newton_explicit_kinetics = false; // the wall-kinetics table feeds the prescribed fluxes instead of the Jacobian
  int cur = 0;  // lapl[cur] = charge row of the current state; lapl[1-cur] = row used by the last step
  // pnp_step with several launches in one call: the batch is cut into row chunks whose launch sequences run on streams of their own
  // (rows are independent), so one chunk's launch boundary is covered by the other chunk's rows; created on first use
  static constexpr int MAX_STEP_STREAMS = 4;
  hipStream_t aux_stream[MAX_STEP_STREAMS - 1] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[MAX_STEP_STREAMS - 1] = {nullptr, nullptr, nullptr};
  int64_t dev_bytes = 0;
  std::string err;
};

static thread_local std::string g_create_error;

// ---- options: key = the name of the environment variable without its CATINT_ prefix (case-insensitive; the prefix is accepted) ----
static bool option_assign(Options& o, const char* key_in, const char* value) {
  if (!key_in || !value) return false;
  std::string key(key_in);
  for (char& ch : key) ch = (char)toupper((unsigned char)ch);
  if (key.rfind("CATINT_", 0) == 0) key = key.substr(7);
  const int iv = atoi(value);
  if (key == "NEWTON_KERNEL") {
    std::string v(value);
    for (char& ch : v) ch = (char)tolower((unsigned char)ch);
    if (v.empty() || v == "auto") o.newton_kernel = NK_AUTO;
    else if (v == "lane2") o.newton_kernel = NK_LANE2;
    else if (v == "lane4") o.newton_kernel = NK_LANE4;
    else if (v[0] == 'l') o.newton_kernel = NK_LANE;
    else if (v[0] == 'g') o.newton_kernel = NK_GENERIC;
    else if (v[0] == 't') o.newton_kernel = NK_TEAM;
    else if (v[0] == 's') o.newton_kernel = NK_SWEEP;
    else if (v[0] == 'b') o.newton_kernel = NK_BOTH;
    else if (v[0] == 'w') o.newton_kernel = NK_WORKGROUP;
    else return false;
  } else if (key == "NEWTON_EXCHANGE") o.newton_exchange_global = (value[0] == 'g' || value[0] == 'G') ? 1 : 0;
  else if (key == "NEWTON_TEAM_THREADS") o.newton_team_threads = iv;
  else if (key == "NEWTON_REGS") o.newton_regs = iv;
  else if (key == "NEWTON_BLOCKS") o.newton_blocks = iv;
  else if (key == "NEWTON_LANE_GROUPS") o.newton_lane_groups = iv;
  else if (key == "NEWTON_SWEEP_BLOCKS") o.newton_sweep_blocks = iv;
  else if (key == "LANE_PIVOT_LIMIT") o.lane_pivot_limit = atof(value);
  else if (key == "LANE_ORDER") o.lane_order = iv;
  else if (key == "LANE_STAGGER") o.lane_stagger = iv;
  else if (key == "LANE_RECORDS") o.lane_records_f32 = (value[0] == 'f' && value[1] == '3') ? 1 : 0;
  else if (key == "LANE_FUSED") o.lane_fused = value[0] ? (iv != 0 ? 1 : 0) : -1;
  else if (key == "PNP_KERNEL") o.pnp_kernel = iv;
  else if (key == "PNP_WAVES_PER_GRID") o.pnp_waves_per_grid = iv;
  else if (key == "PNP_SPECIES_PER_WAVE") o.pnp_species_per_wave = iv;
  else if (key == "PNP_STEP_STREAMS") o.pnp_step_streams = iv;
  else if (key == "PNP_ALTERNATE_ROWS") o.pnp_alternate_rows = value[0] ? (iv != 0 ? 1 : 0) : -1;
  else if (key == "PNP_ST_WAVES_PER_CU") o.pnp_st_waves_per_cu = iv;
  else if (key == "PNP_NO_POST_UPLOAD_DISPATCH") o.pnp_no_post_upload_dispatch = iv != 0 || !value[0] ? 1 : 0;
  else return false;
  return true;
}

static const char* const kOptionKeys[] = {"NEWTON_KERNEL", "NEWTON_EXCHANGE", "NEWTON_TEAM_THREADS", "NEWTON_REGS", "NEWTON_BLOCKS",
                                          "NEWTON_LANE_GROUPS", "NEWTON_SWEEP_BLOCKS", "LANE_PIVOT_LIMIT", "LANE_ORDER", "LANE_STAGGER", "LANE_FUSED", "LANE_RECORDS", "PNP_KERNEL",
                                          "PNP_WAVES_PER_GRID", "PNP_SPECIES_PER_WAVE", "PNP_STEP_STREAMS", "PNP_ALTERNATE_ROWS",
                                          "PNP_ST_WAVES_PER_CU", "PNP_NO_POST_UPLOAD_DISPATCH"};

// the defaults of a new handle: the environment, read once per pnp_create (the only place the library looks at it)
static Options options_from_environment() {
  Options o;
  for (const char* k : kOptionKeys) {
    const std::string name = std::string("CATINT_") + k;
    if (const char* e = getenv(name.c_str())) (void)option_assign(o, k, e);
  }
  return o;
}

static int fail(pnp_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  else g_create_error = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                                      \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(h, e_ == hipErrorOutOfMemory ? PNP_ENOMEM : PNP_EDEVICE,                    \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                         \
  } while (0)

template <typename T>
static hipError_t dev_alloc(pnp_handle* h, T** p, size_t count) {
  hipError_t e = hipMalloc((void**)p, count * sizeof(T));
  if (e == hipSuccess) h->dev_bytes += (int64_t)(count * sizeof(T));
  return e;
}

extern "C" {

const char* pnp_version(void) { return "catint_pnp 0.1 (gfx950)"; }

const char* pnp_last_error(const pnp_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void pnp_destroy(pnp_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void* p : {(void*)h->c, (void*)h->lapl[0], (void*)h->lapl[1], (void*)h->v, (void*)h->gradv, (void*)h->rates,
                  (void*)h->pb, (void*)h->vzeta, (void*)h->flux, (void*)h->cbulk, (void*)h->csurf, (void*)h->status,
                  (void*)h->spec, (void*)h->ytmp, (void*)h->ftmp, (void*)h->c_old, (void*)h->work, (void*)h->iters, (void*)h->stash, (void*)h->rt_dev, (void*)h->rs_dev, (void*)h->c_old2, (void*)h->phi_old2, (void*)h->vol_dev, (void*)h->bdf_acc, (void*)h->wk_k, (void*)h->gw, (void*)h->gv, (void*)h->mol_lapl, (void*)h->scf_d, (void*)h->scf_i, (void*)h->scf_snap, (void*)h->stage, (void*)h->sweep, (void*)h->lane_buf, (void*)h->lane2_buf, (void*)h->lane4_buf, (void*)h->lane_perm, (void*)h->user_mask, (void*)h->ode_buf, (void*)h->ode_int, (void*)h->rkc_d, (void*)h->rkc_i})
    if (p) (void)hipFree(p);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  for (int i = 0; i < pnp_handle::MAX_STEP_STREAMS - 1; ++i) {
    if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    if (h->aux_stream[i]) {
      (void)hipStreamSynchronize(h->aux_stream[i]);
      (void)hipStreamDestroy(h->aux_stream[i]);
    }
  }
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int pnp_create(const pnp_config* cfg, pnp_handle** out) {
  if (!cfg || !out) return fail(nullptr, PNP_EINVAL, "pnp_create: null argument");
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(pnp_config))
    return fail(nullptr, PNP_EINVAL, "pnp_create: pnp_config.struct_size mismatch (ABI)");
  if (cfg->nspecies < 1 || cfg->nspecies > PNP_MAX_SPECIES)
    return fail(nullptr, PNP_EINVAL, "pnp_create: nspecies out of range [1,16]");
  const bool newton = cfg->method == PNP_METHOD_NEWTON;
  const int P = newton ? 1 : points_per_lane(cfg->nx);
  if (P == 0)
    return fail(nullptr, PNP_EINVAL, "pnp_create: nx must be in [5, 4098] (<=16 points per lane, <=4 waves per system)");
  if (newton && (cfg->nx < 3 || cfg->nx > 8192)) return fail(nullptr, PNP_EINVAL, "pnp_create: physical mode needs nx in [3, 8192]");
  if (newton && cfg->nspecies > PNP_NEWTON_MAX_SPECIES)
    return fail(nullptr, PNP_EINVAL, "pnp_create: physical mode supports at most 8 species");
  if (newton && cfg->pb_mode != PNP_PB_DD)
    return fail(nullptr, PNP_EINVAL, "pnp_create: physical mode takes the wall and bulk potentials (PNP_PB_DD)");
  if (cfg->method != PNP_METHOD_CRANK_NICOLSON && cfg->method != PNP_METHOD_FTCS && !newton)
    return fail(nullptr, PNP_EINVAL, "pnp_create: no calculator found with this method");  // calculator_old.py:109-111
  if (cfg->pb_mode < PNP_PB_DD || cfg->pb_mode > PNP_PB_VBULK_GBULK)
    return fail(nullptr, PNP_EINVAL, "pnp_create: unsupported pb_bound combination");
  if (cfg->method == PNP_METHOD_CRANK_NICOLSON && !cfg->use_migration)
    return fail(nullptr, PNP_EINVAL,
                "pnp_create: Crank-Nicolson needs use_migration (the reference reads an unbound 'v', calculator_old.py:514,529)");
  if (cfg->batch_capacity < 1) return fail(nullptr, PNP_EINVAL, "pnp_create: batch_capacity < 1");
  if (!(cfg->dx > 0) || !(cfg->dt > 0) || !(cfg->eps > 0) || !(cfg->beta > 0))
    return fail(nullptr, PNP_EINVAL, "pnp_create: dx, dt, eps, beta must be positive");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(nullptr, PNP_EDEVICE, "pnp_create: no HIP device visible (the transport path has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, PNP_EINVAL, "pnp_create: bad device ordinal");

  pnp_handle* h = new pnp_handle();
  h->opt = options_from_environment();
  h->cfg = *cfg;
  h->P = P;
  h->newton = newton;
  memset(&h->np, 0, sizeof(h->np));
  h->np.struct_size = (int32_t)sizeof(pnp_newton_params);
  h->np.maxit = 50;       // COMSOL maxiter, comsol_model.py:465-516
  h->np.tol = 1e-10;
  h->np.dphi_max = 0.05;
  memset(&h->a, 0, sizeof(DevArgs));
  memset(&h->rt, 0, sizeof(ReactionTable));
  auto bail = [&](int code) {
    g_create_error = h->err;
    pnp_destroy(h);
    return code;
  };
#define HIP_TRYC(expr)                                                                                  \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) {                                                                             \
      h->err = std::string(#expr) + ": " + hipGetErrorString(e_);                                       \
      return bail(e_ == hipErrorOutOfMemory ? PNP_ENOMEM : PNP_EDEVICE);                                \
    }                                                                                                   \
  } while (0)
  HIP_TRYC(hipSetDevice(cfg->device));
  HIP_TRYC(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIP_TRYC(hipEventCreate(&h->ev0));
  HIP_TRYC(hipEventCreate(&h->ev1));
  const int64_t Bc = cfg->batch_capacity;
  const int N = cfg->nspecies;
  const int ldx = (cfg->nx + 15) / 16 * 16;
  HIP_TRYC(dev_alloc(h, &h->c, (size_t)Bc * N * ldx));
  if (!newton) {
    HIP_TRYC(dev_alloc(h, &h->lapl[0], (size_t)Bc * ldx));
    HIP_TRYC(dev_alloc(h, &h->lapl[1], (size_t)Bc * ldx));
  } else {
    HIP_TRYC(dev_alloc(h, &h->c_old, (size_t)Bc * N * ldx));
    HIP_TRYC(dev_alloc(h, &h->v, (size_t)Bc * ldx));
    HIP_TRYC(dev_alloc(h, &h->iters, (size_t)Bc));
    HIP_TRYC(dev_alloc(h, &h->wk_k, (size_t)Bc * PNP_MAX_WALL_REACTIONS));
    HIP_TRYC(dev_alloc(h, &h->gw, (size_t)cfg->nx));
    HIP_TRYC(dev_alloc(h, &h->gv, (size_t)cfg->nx));
    {
      std::vector<double> x((size_t)cfg->nx);
      for (int i = 0; i < cfg->nx; ++i) x[i] = i * cfg->dx;
      h->xgrid = x;
      std::vector<double> w((size_t)cfg->nx, 1.0), v((size_t)cfg->nx, 1.0);
      v[0] = v[cfg->nx - 1] = 0.5;
      HIP_TRYC(hipMemcpy(h->gw, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
      HIP_TRYC(hipMemcpy(h->gv, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    const int nb = N + 1;
    if (!newton_exchange_in_lds(nb, cfg->nx, h->opt)) {
      size_t slice = newton_exchange_doubles(nb, cfg->nx);
      if (nb >= 3 && newton_team_doubles(nb, cfg->nx) > slice) slice = newton_team_doubles(nb, cfg->nx);
      h->work_stride = (int64_t)slice;
      int64_t blocks = Bc < 1024 ? Bc : 1024;
      const int64_t cap = (int64_t)((size_t)4 << 30) / (int64_t)(slice * sizeof(double));
      if (blocks > cap) blocks = cap < 1 ? 1 : cap;
      h->nw_blocks = (int)blocks;
      HIP_TRYC(dev_alloc(h, &h->work, slice * (size_t)blocks));
    } else {
      h->nw_blocks = (int)(Bc < 4096 ? Bc : 4096);
    }
    if (const int ts = newton_pair_stride(nb, cfg->nx)) {
      h->stash_stride = (int64_t)(2 * nb * nb + nb) * ts;
      HIP_TRYC(dev_alloc(h, &h->stash, (size_t)h->stash_stride * (size_t)h->nw_blocks));
    }
  }
  HIP_TRYC(dev_alloc(h, &h->pb, (size_t)Bc * 4));
  HIP_TRYC(dev_alloc(h, &h->vzeta, (size_t)Bc));
  HIP_TRYC(dev_alloc(h, &h->flux, (size_t)Bc * N));
  HIP_TRYC(dev_alloc(h, &h->cbulk, (size_t)Bc * N));
  HIP_TRYC(dev_alloc(h, &h->csurf, (size_t)Bc * N));
  HIP_TRYC(dev_alloc(h, &h->status, (size_t)Bc));
  HIP_TRYC(dev_alloc(h, &h->spec, (size_t)PNP_MAX_SPECIES));
#undef HIP_TRYC
  DevArgs& a = h->a;
  a.N = N;
  a.nx = cfg->nx;
  a.m = cfg->nx - 2;
  a.ldx = ldx;
  a.pb_mode = cfg->pb_mode;
  a.method = cfg->method;
  a.lf = cfg->lax_friedrich ? 1 : 0;
  a.use_mig = cfg->use_migration ? 1 : 0;
  a.nsteps = 1;
  a.has_rates = 0;
  a.st_waves_per_cu = h->opt.pnp_st_waves_per_cu;
  a.B = 0;
  a.dx = cfg->dx;
  a.dx2 = cfg->dx * cfg->dx;
  a.inv2dx = 1.0 / (2 * cfg->dx);
  a.nxm1 = (double)(cfg->nx - 1);
  a.dt = cfg->dt;
  a.beta = cfg->beta;
  a.eps = cfg->eps;
  a.c = h->c;
  a.pb = h->pb;
  a.vzeta = h->vzeta;
  a.flux = h->flux;
  a.cbulk = h->cbulk;
  a.status = h->status;
  a.spec = h->spec;
  a.rates = nullptr;
  *out = h;
  return PNP_OK;
}

int pnp_set_option(pnp_handle* h, const char* key, const char* value) {
  if (!h || !key || !value) return fail(h, PNP_EINVAL, "pnp_set_option: null argument");
  Options o = h->opt;
  if (!option_assign(o, key, value)) return fail(h, PNP_EINVAL, std::string("pnp_set_option: unknown key or value: ") + key + " = " + value);
  if (o.newton_exchange_global != h->opt.newton_exchange_global)
    return fail(h, PNP_ESTATE, "pnp_set_option: NEWTON_EXCHANGE sizes the buffers of pnp_create; set CATINT_NEWTON_EXCHANGE before creating the handle");
  // workspaces sized by an option are allocated on first use: an option set afterwards must not outgrow them
  if ((h->lane_buf || h->lane2_buf || h->lane4_buf) && o.newton_lane_groups != h->opt.newton_lane_groups)
    return fail(h, PNP_ESTATE, "pnp_set_option: NEWTON_LANE_GROUPS after the lane workspace was allocated");
  if (h->sweep && o.newton_sweep_blocks != h->opt.newton_sweep_blocks)
    return fail(h, PNP_ESTATE, "pnp_set_option: NEWTON_SWEEP_BLOCKS after the sweep workspace was allocated");
  h->opt = o;
  h->a.st_waves_per_cu = o.pnp_st_waves_per_cu;
  return PNP_OK;
}

int64_t pnp_get_lane_order(pnp_handle* h, int32_t* perm) {
  if (!h) return PNP_EINVAL;
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_get_lane_order: the handle was not created with PNP_METHOD_NEWTON");
  const int64_t n = h->lane_perm_B;
  if ((int64_t)h->lane_perm_host.size() < n) return 0;
  if (perm)
    for (int64_t s_ = 0; s_ < n; ++s_) perm[s_] = h->lane_perm_host[(size_t)s_];
  return n;
}

int pnp_set_species(pnp_handle* h, const double* D, const double* charges) {
  if (!h || !D || !charges) return fail(h, PNP_EINVAL, "pnp_set_species: null argument");
  SpecConst sc[PNP_MAX_SPECIES];
  memset(sc, 0, sizeof(sc));
  const double dx = h->a.dx, dt = h->a.dt, beta = h->a.beta, eps = h->a.eps;
  for (int k = 0; k < h->a.N; ++k) {
    if (!(D[k] >= 0) || !std::isfinite(charges[k])) return fail(h, PNP_EINVAL, "pnp_set_species: bad D or charge");
    if (h->newton) {
      if (!(D[k] > 0)) return fail(h, PNP_EINVAL, "pnp_set_species: physical mode needs D > 0");
      h->Dk[k] = D[k];
      h->qk[k] = charges[k];
    }
    SpecConst& S = sc[k];
    S.D = D[k];
    S.q = charges[k];
    S.mu = D[k] * charges[k] * beta;                 // transport.py:436
    S.sf = D[k] * dt / (dx * dx);                    // calculator_old.py:543 / :1013
    S.s = S.sf;
    if (h->a.lf) S.s += 0.5;                         // :544-546
    S.hs = 0.5 * S.s;
    S.ee = charges[k] * beta * dt * D[k];            // :547
    S.e4 = S.ee / 4. / dx;
    S.rdiag = 1.0 / (1.0 + S.s);
    S.oms = 1 - S.s;
    S.hsr = S.hs * S.rdiag;
    S.e4r = S.e4 * S.rdiag;
    S.eer = S.ee * S.rdiag;
    S.omsr = S.oms * S.rdiag;
    S.qe = charges[k] / eps;
    S.twoD = 2 * D[k];
    S.dm = dt / (2. * dx) * S.mu;                    // :1014
    S.Mf = -2. * D[k] * dt / (dx * dx);              // :1015
    if (!h->a.lf) S.Mf += 1;                         // :1021
  }
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(h->spec, sc, sizeof(sc), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->have_species = true;
  return PNP_OK;
}

int pnp_set_reactions(pnp_handle* h, int32_t nreactions, const int32_t* n_lhs, const int32_t* lhs, const int32_t* n_rhs,
                      const int32_t* rhs, const double* kf, const double* kr) {
  if (!h) return PNP_EINVAL;
  if (nreactions < 0 || nreactions > PNP_MAX_REACTIONS) return fail(h, PNP_EINVAL, "pnp_set_reactions: too many reactions");
  if (nreactions > 0 && h->cfg.method != PNP_METHOD_FTCS && !h->newton)
    return fail(h, PNP_EINVAL, "pnp_set_reactions: only FTCS has a rate term (calculator_old.py:1022; CN has none)");
  ReactionTable& rt = h->rt;
  memset(&rt, 0, sizeof(rt));
  rt.n = nreactions;
  for (int r = 0; r < nreactions; ++r) {
    if (n_lhs[r] < 0 || n_lhs[r] > PNP_MAX_REACTANTS || n_rhs[r] < 0 || n_rhs[r] > PNP_MAX_REACTANTS)
      return fail(h, PNP_EINVAL, "pnp_set_reactions: too many reactants");
    rt.n_lhs[r] = n_lhs[r];
    rt.n_rhs[r] = n_rhs[r];
    for (int j = 0; j < n_lhs[r]; ++j) {
      const int k = lhs[r * PNP_MAX_REACTANTS + j];
      if (k < 0 || k >= h->a.N) return fail(h, PNP_EINVAL, "pnp_set_reactions: species index out of range");
      rt.lhs[r][j] = k;
    }
    for (int j = 0; j < n_rhs[r]; ++j) {
      const int k = rhs[r * PNP_MAX_REACTANTS + j];
      if (k < 0 || k >= h->a.N) return fail(h, PNP_EINVAL, "pnp_set_reactions: species index out of range");
      rt.rhs[r][j] = k;
    }
    rt.kf[r] = kf[r];
    rt.kr[r] = kr[r];
  }
  if (h->newton) {
    // physical mode: mass action in activities with every reaction summed (comsol_model.py:781-867); the table lives
    // in device memory and is read by the assembly
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!h->rt_dev) HIP_TRY(h, dev_alloc(h, &h->rt_dev, 1));
    HIP_TRY(h, hipMemcpyAsync(h->rt_dev, &rt, sizeof(rt), hipMemcpyHostToDevice, h->stream));
    // the same table per reaction side, for the lane kernels (see ReactionSides)
    static ReactionSides rs;      // (host staging; copied before this call returns)
    memset(&rs, 0, sizeof(rs));
    const int N = h->a.N;
    for (int r = 0; r < nreactions; ++r) {
      for (int sd = 0; sd < 2; ++sd) {
        const double kk = sd == 0 ? rt.kf[r] : rt.kr[r];
        if (kk == 0.0) continue;
        ReactionSides::Side& S = rs.side[rs.n++];
        S.k = kk;
        const int n = sd == 0 ? rt.n_lhs[r] : rt.n_rhs[r];
        const int32_t* idx = sd == 0 ? rt.lhs[r] : rt.rhs[r];
        S.order = n;
        int cnt[PNP_MAX_SPECIES] = {0};
        for (int a = 0; a < n; ++a) cnt[idx[a]] += 1;
        for (int k = 0; k < N && k < PNP_NEWTON_MAX_SPECIES; ++k)
          if (cnt[k] > rs.max_exponent) rs.max_exponent = cnt[k];
        for (int a = 0; a < PNP_MAX_REACTANTS; ++a) {
          const uint32_t row = a < n ? (uint32_t)idx[a] : 8u, col = a < n ? (uint32_t)idx[a] : 15u;
          S.slots |= (row << (4 * a)) | (col << (16 + 4 * a));
        }
        const double sg = sd == 0 ? 1.0 : -1.0;                    // forward minus backward
        for (int a = 0; a < rt.n_lhs[r]; ++a) S.w[rt.lhs[r][a]] -= sg;   // educts lose
        for (int a = 0; a < rt.n_rhs[r]; ++a) S.w[rt.rhs[r][a]] += sg;   // products gain
      }
    }
    if (rs.n & 1) {       // the kernels walk the sides two at a time (independent dependency chains): pad with a side without a rate
      ReactionSides::Side& S = rs.side[rs.n++];
      for (int a = 0; a < PNP_MAX_REACTANTS; ++a) S.slots |= (8u << (4 * a)) | (15u << (16 + 4 * a));
    }
    h->rs_max_exponent = rs.max_exponent;
    if (!h->rs_dev) HIP_TRY(h, dev_alloc(h, &h->rs_dev, 1));
    HIP_TRY(h, hipMemcpyAsync(h->rs_dev, &rs, sizeof(rs), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PNP_OK;
  }
  if (nreactions > 0 && !h->rates) {
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, dev_alloc(h, &h->rates, (size_t)h->cfg.batch_capacity * h->a.N * h->a.ldx));
    HIP_TRY(h, hipMemsetAsync(h->rates, 0, (size_t)h->cfg.batch_capacity * h->a.N * h->a.ldx * sizeof(double), h->stream));
  }
  h->a.has_rates = nreactions > 0 ? 1 : 0;
  h->a.rates = h->rates;
  return PNP_OK;
}

int pnp_set_flux(pnp_handle* h, const double* flux) {
  if (!h || !flux) return fail(h, PNP_EINVAL, "pnp_set_flux: null argument");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_set_flux: call pnp_set_batch first");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(h->flux, flux, (size_t)h->B * h->a.N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return PNP_OK;
}

int pnp_set_pb(pnp_handle* h, const double* pb, const double* vzeta) {
  if (!h || !pb || !vzeta) return fail(h, PNP_EINVAL, "pnp_set_pb: null argument");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_set_pb: call pnp_set_batch first");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(h->pb, pb, (size_t)h->B * 4 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->vzeta, vzeta, (size_t)h->B * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->newton) {
    h->lane_key.resize((size_t)h->B);
    for (int64_t b = 0; b < h->B; ++b) h->lane_key[b] = (float)std::fabs(pb[b * 4 + 0] - pb[b * 4 + 1]);
  }
  return PNP_OK;
}

static int ensure_potential_buffers(pnp_handle* h);

int pnp_set_batch(pnp_handle* h, int64_t B, const double* c0, const double* pb, const double* vzeta, const double* flux) {
  if (!h || !c0 || !pb || !vzeta || !flux) return fail(h, PNP_EINVAL, "pnp_set_batch: null argument");
  if (!h->have_species) return fail(h, PNP_ESTATE, "pnp_set_batch: call pnp_set_species first");
  if (B < 1 || B > h->cfg.batch_capacity) return fail(h, PNP_EINVAL, "pnp_set_batch: B outside [1, batch_capacity]");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const int N = h->a.N, nx = h->a.nx, ldx = h->a.ldx;
  h->B = B;
  h->newton_mask = nullptr;      // (a lane mask belongs to the batch it was set for)
  h->a.B = B;
  // One contiguous upload into a staging buffer; unpack_state_kernel writes the pitched rows (zero pads: they travel through
  // the kernels untouched), the bulk Dirichlet values = last grid point of the initial state (calculator_old.py:540) and the
  // zeroed second charge row.
  if (!h->stage) HIP_TRY(h, dev_alloc(h, &h->stage, (size_t)h->cfg.batch_capacity * N * nx));
  HIP_TRY(h, hipMemcpyAsync(h->stage, c0, (size_t)B * N * nx * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->pb, pb, (size_t)B * 4 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->vzeta, vzeta, (size_t)B * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->flux, flux, (size_t)B * N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, launch_unpack_state(h->a, h->stage, h->cbulk, h->newton ? nullptr : h->lapl[1], h->newton ? h->iters : nullptr, h->stream));
  if (h->newton) {
    // initial guess of the potential: the bulk value everywhere (field-free electrolyte, c = c_bulk; comsol_model.py:744)
    std::vector<double> phi0((size_t)B * ldx, 0.0);
    for (int64_t b = 0; b < B; ++b)
      for (int i = 0; i < nx; ++i) phi0[(size_t)b * ldx + i] = pb[b * 4 + 1];
    HIP_TRY(h, hipMemcpyAsync(h->v, phi0.data(), phi0.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->have_batch = true;
    h->steps_done = 0;
    h->bdf_history = false;
    // lane kernels: the iteration counters were just zeroed (unpack_state_kernel), so the first solve of this batch deals the
    // operating points by their wall-to-bulk potential difference
    h->iters_valid = false;
    h->lane_key.resize((size_t)B);
    for (int64_t b = 0; b < B; ++b) h->lane_key[b] = (float)std::fabs(pb[b * 4 + 0] - pb[b * 4 + 1]);
    return PNP_OK;
  }
  h->cur = 0;
  HIP_TRY(h, launch_charge_row(h->a, h->lapl[0], h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  // Potential and gradient of the uploaded state (what pnp_get_state would compute on demand).  It is also the dispatch that
  // keeps the first large launch after an upload at the sustained rate: measured on MI355X (profiles/r02_slow_start_after_upload.txt,
  // DESIGN.md section 6), a launch that fills every SIMD in one round runs 1.45 x slower for its whole duration -- and so do the
  // next one or two -- when the dispatch before it ON THIS QUEUE was one of the upload's small element-wise kernels; after a
  // wave-per-operating-point kernel (this one, or a one-step launch) it runs at full rate.  Dispatches on other queues, idle time
  // and extra synchronisation change nothing.
  // Only batches that can fill every SIMD in one round were affected (B = 960, 1024 with the headline kernel; B = 512 was not): smaller
  // uploads -- the per-iteration uploads of a small compat SCF loop -- skip the extra dispatch, its buffers and its synchronisation.
  if (h->B >= 768 && !h->opt.pnp_no_post_upload_dispatch) {
    const int rc = ensure_potential_buffers(h);
    if (rc != PNP_OK) return rc;
    HIP_TRY(h, launch_poisson(h->a, h->lapl[0], h->v, h->gradv, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  h->have_batch = true;
  h->steps_done = 0;
  return PNP_OK;
}

// nearest measured entry: exact points per lane and launch mode, nearest species count (ties to the smaller), nearest batch
// bucket on a log scale
static const StepTableEntry* lookup_step(int N, int P, int64_t B, bool fused) {
  const StepTableEntry* best = nullptr;
  double best_d = 1e300;
  for (int i = 0; i < kStepTableSize; ++i) {
    const StepTableEntry& e = kStepTable[i];
    if (e.P != P || e.fused != (fused ? 1 : 0)) continue;
    const double dn = std::fabs((double)(e.N - N)) + (e.N > N ? 0.25 : 0.0);
    const double db = std::fabs(std::log2((double)(B < 1 ? 1 : B)) - std::log2((double)e.B));
    const double d = dn * 100.0 + db;
    if (d < best_d) {
      best_d = d;
      best = &e;
    }
  }
  return best;
}

// the rows [b0, b0 + nb) of the batch as a batch of their own
static DevArgs row_chunk(const DevArgs& a, int64_t b0, int64_t nb) {
  DevArgs r = a;
  r.B = nb;
  r.c = a.c + (size_t)b0 * a.N * a.ldx;
  r.lapl_a = a.lapl_a + (size_t)b0 * a.ldx;
  r.lapl_b = a.lapl_b + (size_t)b0 * a.ldx;
  if (a.rates) r.rates = a.rates + (size_t)b0 * a.N * a.ldx;
  r.pb = a.pb + (size_t)b0 * 4;
  r.vzeta = a.vzeta + b0;
  r.flux = a.flux + (size_t)b0 * a.N;
  r.cbulk = a.cbulk + (size_t)b0 * a.N;
  r.status = a.status + b0;
  return r;
}

// one or more fused timesteps in a single launch over the rows [b0, b0 + nb) on `stream`; Bsel: the batch size the kernel
// choice is made for (the whole call's, so that every chunk of a call runs the same kernel)
static int run_steps_on(pnp_handle* h, int nsteps, int64_t b0, int64_t nb, hipStream_t stream) {
  DevArgs a = h->a;
  const int64_t Bsel = a.B;
  a.nsteps = nsteps;
  a.lapl_a = h->lapl[h->cur];
  a.lapl_b = h->lapl[1 - h->cur];
  // One timestep per launch over a state the caches cannot hold: every other launch walks the operating points from the last to
  // the first, starting on the rows the launch before wrote last (still in L2 / MALL).  Measured (tools/probe/step_streams.py,
  // profiles/r03_step_streams.jsonl): +6 % at 335 MB of state, nothing lost or gained at 2.1 GB, -2 % at 42 MB.
  const bool alternate = h->opt.pnp_alternate_rows >= 0 ? h->opt.pnp_alternate_rows == 1 : (double)a.B * (a.N + 2) * a.ldx * 8.0 >= 128e6;
  a.reverse = (nsteps == 1 && alternate) ? (int)(h->steps_done & 1) : 0;
  if (a.has_rates) a.rates = h->rates;
  if (b0 != 0 || nb != a.B) a = row_chunk(a, b0, nb);
  if (a.has_rates) {
    if (nsteps != 1) return fail(h, PNP_EINVAL, "internal: rate terms need one step per launch");
    HIP_TRY(h, launch_rates(a, h->rt, const_cast<double*>(a.rates), stream));
  }
  // grids beyond 1026 points: several waves per tridiagonal system
  if (waves_per_system(a.nx) > 1) {
    HIP_TRY(h, launch_step_mw(a, stream));
    return PNP_OK;
  }
  // Which kernel: the measured table (pnp_step_table.h, generated by tools/make_step_table.py from a sweep of every variant
  // over species count x points per lane x batch x fused/per-step on an MI355X: profiles/r02_config_sweep.jsonl) at the nearest
  // measured species count and batch bucket.  The register-resident and streaming kernels cover the Dirichlet/Dirichlet Poisson
  // branch with an even number of points per lane and no FTCS rate term; everything else runs the LDS-staged kernel with the
  // table's (W, G) when the entry is of that family, else with choose_step_config's.
  const bool fused = nsteps >= 8;
  const StepTableEntry* te = lookup_step(a.N, h->P, Bsel, fused);
  int kind = te ? te->kind : 0, W = te ? te->W : 1, G = te ? te->G : 1;
  const bool direct_ok = step_rr_applicable(a);
  if (kind != 0 && !direct_ok) kind = 0, W = 0;
  if (h->opt.pnp_kernel == 2) kind = 0, W = (te && te->kind == 0) ? W : 0;
  if (h->opt.pnp_kernel == 4 && direct_ok) kind = 1, W = (te && te->kind == 1) ? W : 1;
  if (h->opt.pnp_kernel >= 5 && h->opt.pnp_kernel <= 7 && direct_ok) kind = 2;
  if (kind == 2) {
    // 16 points per lane: the gradient row of the step in LDS (two waves per SIMD instead of one: 0.53 -> 0.60 of the roofline per
    // step and 0.68 -> 0.72 fused on one GPU's share of configs[3]); shorter grids: registers only
    const int st_mode = h->opt.pnp_kernel == 6 ? 1 : (h->opt.pnp_kernel == 7 ? 2 : (h->opt.pnp_kernel == 5 ? 0 : ((h->P == 16 && Bsel >= 2048) ? 2 : 0)));
    HIP_TRY(h, launch_step_st(a, st_mode, stream));
  } else if (kind == 1) {
    int w = W;
    if (h->opt.pnp_waves_per_grid >= 1 && h->opt.pnp_waves_per_grid <= 4) w = h->opt.pnp_waves_per_grid;
    HIP_TRY(h, launch_step_rr(a, w, stream));
  } else {
    if (W == 0 || !step_config_supported(W, G)) choose_step_config(a.N, Bsel, h->P, fused, &W, &G);
    if (h->opt.pnp_waves_per_grid >= 1 || h->opt.pnp_species_per_wave >= 1) {
      const int w2 = h->opt.pnp_waves_per_grid >= 1 ? h->opt.pnp_waves_per_grid : W;
      const int g2 = h->opt.pnp_species_per_wave >= 1 ? h->opt.pnp_species_per_wave : 1;
      if (step_config_supported(w2, g2)) {
        W = w2;
        G = g2;
      }
    }
    HIP_TRY(h, launch_step(a, W, G, stream));
  }
  return PNP_OK;
}

static int run_steps(pnp_handle* h, int nsteps) {
  const int rc = run_steps_on(h, nsteps, 0, h->a.B, h->stream);
  if (rc != PNP_OK) return rc;
  if (nsteps & 1) h->cur = 1 - h->cur;
  h->steps_done += nsteps;
  return PNP_OK;
}

// Row chunks for a pnp_step call of `launches` launches: 1 = the whole batch on the handle's stream.
// Measured (tools/probe/step_streams.py, profiles/r03_step_streams.jsonl): two chunks +3 ... 4 % from 4096 operating points up
// (one chunk's launch boundary is covered by the other's rows), three and four no better than two, and a loss below that (a
// chunk no longer fills the chip: -4 % at 2048, -11 % at 1024).
static int step_streams(const pnp_handle* h, int launches) {
  if (launches < 2) return 1;
  int S = h->opt.pnp_step_streams;
  if (S < 1) return h->a.B >= 4096 ? 2 : 1;
  if (S > pnp_handle::MAX_STEP_STREAMS) S = pnp_handle::MAX_STEP_STREAMS;
  while (S > 1 && h->a.B / S < 256) --S;
  return S;
}

// Lane kernels: a wave iterates until the slowest of its operating points is done, and the points that are done keep streaming their
// records (pnp_lane.hip) -- 1.15-1.28 x the algorithmic HBM bytes measured on random batches.  So the operating points are dealt to
// the slots (group, lane) in the order of the Newton iterations they are expected to need, most first (the long waves start first,
// the short ones fill the tail): by the iteration counts of the previous Newton call on this batch (read back here: one small copy
// and a counting sort per call), before that by the wall-to-bulk potential difference.  The arithmetic of a point does not depend on
// its slot: results are the same to the bit with and without the order (tests/test_gpu_lane.py).
static int lane_order(pnp_handle* h, NewtonArgs& a) {
  a.lane_perm = nullptr;
  const int64_t B = h->B;
  if (!h->lane_perm_keep) h->lane_perm_B = 0;
  // a solve restricted by pnp_set_lane_mask (the rerun ladder's confirming solve: a handful of recovered lanes): only the lanes it solves
  // are dealt to slots, and the launch covers ceil(n / points per group) groups instead of the whole batch
  const bool masked = h->newton_mask && h->newton_mask == h->user_mask && (int64_t)h->user_mask_host.size() == B && h->user_mask_count < B;
  if ((h->opt.lane_order == 0 || B < 64) && !masked) return PNP_OK;
  if (h->lane_perm_keep && h->lane_perm && h->lane_perm_B > 0) {      // the order of this call's first launch
    a.lane_perm = h->lane_perm;
    a.B = h->lane_perm_B;
    return PNP_OK;
  }
  if (!h->lane_perm) HIP_TRY(h, dev_alloc(h, &h->lane_perm, (size_t)h->cfg.batch_capacity));
  std::vector<int32_t>& perm = h->lane_perm_host;
  perm.resize((size_t)B);
  if (h->iters_valid && h->opt.lane_order != 0 && B >= 64) {
    std::vector<int32_t>& it = h->lane_iters_host;
    it.resize((size_t)B);
    HIP_TRY(h, hipMemcpyAsync(it.data(), h->iters, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    constexpr int KMAX = 4096;
    std::vector<int64_t> start(KMAX + 1, 0);
    for (int64_t b = 0; b < B; ++b) {
      const int k = it[b] < 0 ? 0 : (it[b] >= KMAX ? KMAX - 1 : it[b]);
      start[KMAX - 1 - k + 1] += 1;                      // descending
    }
    for (int k = 0; k < KMAX; ++k) start[k + 1] += start[k];
    for (int64_t b = 0; b < B; ++b) {
      const int k = it[b] < 0 ? 0 : (it[b] >= KMAX ? KMAX - 1 : it[b]);
      perm[(size_t)start[KMAX - 1 - k]++] = (int32_t)b;
    }
  } else if ((int64_t)h->lane_key.size() == B && h->opt.lane_order != 0 && B >= 64) {
    for (int64_t b = 0; b < B; ++b) perm[(size_t)b] = (int32_t)b;
    const std::vector<float>& key = h->lane_key;
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t x, int32_t y) { return key[(size_t)x] > key[(size_t)y]; });
  } else if (masked) {
    for (int64_t b = 0; b < B; ++b) perm[(size_t)b] = (int32_t)b;
  } else {
    return PNP_OK;
  }
  if (masked) {      // keep the order, drop the lanes that are not solved
    size_t n = 0;
    for (int64_t s_ = 0; s_ < B; ++s_)
      if (h->user_mask_host[(size_t)perm[(size_t)s_]] != 0) perm[n++] = perm[(size_t)s_];
    for (size_t s_ = n; s_ < (size_t)B; ++s_) perm[s_] = perm[n > 0 ? n - 1 : 0];
    a.B = (int64_t)n;
  }
  HIP_TRY(h, hipMemcpyAsync(h->lane_perm, perm.data(), (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));        // (the host vector may be rewritten by the next call)
  a.lane_perm = h->lane_perm;
  h->lane_perm_B = a.B;
  return PNP_OK;
}

// ---- BDF2 and the predictor (pnp_newton_params.time_order = 2 / .predictor = 1; the reference's transient study asks COMSOL for its
// BDF time stepper with maxorder 2, comsol_model.py:518-531)
// (3 c_n+1 - 4 c_n + c_n-1) / (2 dt) = (3/2) (c_n+1 - c*) / dt with c* = (4 c_n - c_n-1) / 3: a backward-Euler step against the
// combination c* with 1/dt scaled by 3/2.  Predictor: the Newton iteration of step n+1 starts from the linear extrapolation
// 2 u_n - u_n-1 of the two previous levels (concentrations and potential) instead of from u_n -- the start of a BDF stepper; same
// equations, same stopping rule, fewer iterations.  A concentration is not extrapolated below a tenth of its value, and a point whose
// extrapolated ions would fill more than 90 % of the volume keeps u_n.  One launch per timestep: the previous-level combination and the
// start are prepared here (one thread per grid point), the kernels take c_old as given.
__global__ void step_prepare_kernel(double* __restrict__ c, double* __restrict__ c2, double* __restrict__ cold, double* __restrict__ phi,
                                    double* __restrict__ phi2, int N, int ldx, int nx, int64_t B, int bdf2, int predictor,
                                    const double* __restrict__ vol /* [N] device copy, or null */) {
  const int64_t total = B * (int64_t)ldx;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = t / ldx;
    const int i = (int)(t - b * ldx);
    if (i >= nx) continue;
    double fill = 0.0;
    if (predictor) {
      for (int k = 0; k < N; ++k) {
        const size_t e = ((size_t)b * N + k) * ldx + i;
        const double cn = c[e], g = 2.0 * cn - c2[e];
        fill += (vol ? vol[k] : 0.0) * (g < 0.1 * cn ? 0.1 * cn : g);
      }
    }
    const bool extrapolate = predictor && fill < 0.9;
    for (int k = 0; k < N; ++k) {
      const size_t e = ((size_t)b * N + k) * ldx + i;
      const double cn = c[e], cm = c2[e];
      cold[e] = bdf2 ? (4.0 * cn - cm) / 3.0 : cn;
      c2[e] = cn;
      if (extrapolate) {
        const double g = 2.0 * cn - cm;
        c[e] = g < 0.1 * cn ? 0.1 * cn : g;
      }
    }
    if (predictor) {
      const size_t e = (size_t)b * ldx + i;
      const double pn = phi[e], pm = phi2[e];
      phi2[e] = pn;
      if (extrapolate) phi[e] = 2.0 * pn - pm;
    }
  }
}
// per-lane bookkeeping over the launches of one call: iteration counts add up, the worst status stays
__global__ void bdf2_accumulate_kernel(int32_t* __restrict__ acc, int32_t* __restrict__ iters, int32_t* __restrict__ status, int64_t B, int last) {
  const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int32_t it = acc[b] + iters[b], st = status[b] > acc[B + b] ? status[b] : acc[B + b];
  acc[b] = it;
  acc[B + b] = st;
  if (last) {
    iters[b] = it;
    status[b] = st;
  }
}

static int run_newton(pnp_handle* h, int nsteps, bool stationary, double tol, int maxit);

// kernel variant of the handle's physics: 0 point ions, 1 steric ions, 2 + homogeneous reactions and / or the constant convection term
// (variant 3 is supported by none of the lane kernels: a guard for tables their flattened form could not hold -- none at present,
// a side has at most PNP_MAX_REACTANTS reactants and the form keeps them one by one)
static int newton_variant(const pnp_handle* h) {
  const bool rt = h->rt_dev && h->rt.n > 0;
  if (rt && h->rs_max_exponent > PNP_MAX_REACTANTS) return 3;
  return (rt || h->velocity != 0.0) ? 2 : (h->mpb ? 1 : 0);
}

// (a solve restricted to a few lanes by pnp_set_lane_mask is sized by those lanes: the workgroup-per-point kernels skip masked-out
// points at once, the lane kernels would walk every group)
static int64_t newton_effective_batch(const pnp_handle* h) {
  const bool host_mask = h->newton_mask && h->newton_mask == h->user_mask && (int64_t)h->user_mask_host.size() == h->B;
  return host_mask ? (h->user_mask_count > 0 ? h->user_mask_count : 1) : h->B;
}

// will run_newton hand this batch to one of the lane kernels (which take BDF2 steps inside one launch)?
static bool newton_lane_family(const pnp_handle* h) {
  const int nb = h->a.N + 1, nx = h->a.nx, variant = newton_variant(h);
  const int64_t n_eff = newton_effective_batch(h);
  return newton_lane4_preferred(nb, nx, n_eff, variant, h->opt) || newton_lane2_preferred(nb, nx, n_eff, variant, h->opt) ||
         newton_lane_preferred(nb, nx, n_eff, variant, h->opt);
}

// nsteps timesteps of the physical mode: one launch (backward Euler), or one launch per step (BDF2 and / or the predictor)
static int newton_timesteps(pnp_handle* h, int nsteps) {
  const bool bdf2 = h->np.time_order == 2, pred = h->np.predictor == 1;
  if ((!bdf2 && !pred) || nsteps < 1) return run_newton(h, nsteps, false, 0.0, 0);
  const int N = h->a.N, ldx = h->a.ldx, nx = h->a.nx;
  const size_t n = (size_t)h->B * N * ldx;
  const int64_t B = h->B;
  const size_t cap = (size_t)h->cfg.batch_capacity;
  if (!h->c_old2) HIP_TRY(h, dev_alloc(h, &h->c_old2, cap * N * ldx));
  if (bdf2 && !pred && newton_lane_family(h)) {
    // the lane kernels keep the history themselves: ONE launch for all nsteps (the first step of a trajectory is backward Euler
    // from u_0, which becomes the history -- as below), c_old2 is the history's home between launches
    h->nw_bdf2_inline = 1;
    h->nw_bdf_hist0 = (h->steps_done == 0 || !h->bdf_history) ? 0 : 1;
    const int rc = run_newton(h, nsteps, false, 0.0, 0);
    h->nw_bdf2_inline = 0;
    h->nw_bdf_hist0 = 0;
    if (rc != PNP_OK) return rc;
    h->bdf_history = true;
    return PNP_OK;
  }
  if (pred && !h->phi_old2) HIP_TRY(h, dev_alloc(h, &h->phi_old2, cap * ldx));
  if (pred && h->mpb && !h->vol_dev) HIP_TRY(h, dev_alloc(h, &h->vol_dev, (size_t)PNP_NEWTON_MAX_SPECIES));
  if (pred && h->mpb) HIP_TRY(h, hipMemcpyAsync(h->vol_dev, h->volk, sizeof(double) * PNP_NEWTON_MAX_SPECIES, hipMemcpyHostToDevice, h->stream));
  if (!h->bdf_acc) HIP_TRY(h, dev_alloc(h, &h->bdf_acc, cap * 2));
  HIP_TRY(h, hipMemsetAsync(h->bdf_acc, 0, (size_t)B * 2 * sizeof(int32_t), h->stream));
  struct KeepPerm {      // (the launches of this call after the first keep its lane order)
    pnp_handle* h;
    ~KeepPerm() { h->lane_perm_keep = false; }
  } keep_guard{h};
  for (int s = 0; s < nsteps; ++s) {
    h->lane_perm_keep = s > 0;
    if (h->steps_done == 0 || !h->bdf_history) {      // first step of a trajectory: backward Euler from u_0, which becomes the history
      HIP_TRY(h, hipMemcpyAsync(h->c_old2, h->c, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
      if (pred) HIP_TRY(h, hipMemcpyAsync(h->phi_old2, h->v, (size_t)B * ldx * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
      const int rc = run_newton(h, 1, false, 0.0, 0);
      if (rc != PNP_OK) return rc;
      h->bdf_history = true;
    } else {
      hipLaunchKernelGGL(step_prepare_kernel, dim3(2048), dim3(256), 0, h->stream, h->c, h->c_old2, h->c_old, h->v, h->phi_old2, N, ldx, nx, B,
                         bdf2 ? 1 : 0, pred ? 1 : 0, (const double*)(pred && h->mpb ? h->vol_dev : nullptr));
      HIP_TRY(h, hipGetLastError());
      h->nw_ext_old = 1;
      h->nw_sig_scale = bdf2 ? 1.5 : 1.0;
      const int rc = run_newton(h, 1, false, 0.0, 0);
      h->nw_ext_old = 0;
      h->nw_sig_scale = 1.0;
      if (rc != PNP_OK) return rc;
    }
    hipLaunchKernelGGL(bdf2_accumulate_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, h->stream, h->bdf_acc, h->iters, h->status, B,
                       s + 1 == nsteps ? 1 : 0);
    HIP_TRY(h, hipGetLastError());
  }
  return PNP_OK;
}

// physical mode: nsteps backward-Euler steps (stationary: one solve with 1/dt = 0) in one launch
static int run_newton(pnp_handle* h, int nsteps, bool stationary, double tol, int maxit) {
  if (stationary) h->bdf_history = false;      // (a stationary solve moves the state off the trajectory)
  NewtonArgs a;
  memset(&a, 0, sizeof(a));
  const int N = h->a.N, nx = h->a.nx;
  const double dx = h->a.dx, dt = h->a.dt, beta = h->a.beta, eps = h->a.eps;
  a.N = N;
  a.nx = nx;
  a.ldx = h->a.ldx;
  a.nsteps = nsteps;
  a.maxit = maxit > 0 ? maxit : h->np.maxit;
  a.wall_bc = h->np.wall_bc;
  a.mpb = h->mpb ? 1 : 0;
  a.estimate = h->np.error_estimate ? 1 : 0;
  a.stationary = stationary ? 1 : 0;
  a.RS = (nx + 15) / 16 * 16;
  a.B = h->B;
  a.work = h->work;
  a.work_stride = h->work_stride ? h->work_stride : (int64_t)newton_exchange_doubles(N + 1, nx);
  a.stash = h->stash;
  a.stash_stride = h->stash_stride;
  a.tol = tol > 0 ? tol : h->np.tol;
  a.dphi_max = h->np.dphi_max;
  a.stern = dx * h->np.stern_capacitance / eps;
  a.phi_pzc = h->np.phi_pzc;
  double qmax = 1.0;
  for (int k = 0; k < N; ++k) {
    a.qb[k] = h->qk[k] * beta;
    a.sig[k] = stationary ? 0.0 : h->nw_sig_scale * (dx * dx / (h->Dk[k] * dt));
    a.fl[k] = dx / h->Dk[k];
    a.peq[k] = dx * dx / eps * h->qk[k];
    a.vol[k] = h->volk[k];
    a.rs[k] = dx * dx / h->Dk[k];
    a.pe[k] = h->velocity * dx / h->Dk[k];
    if (fabs(h->qk[k]) > qmax) qmax = fabs(h->qk[k]);
  }
  a.convect = h->velocity != 0.0 ? 1 : 0;
  a.vt_inv = beta * qmax;
  a.rt = (h->rt_dev && h->rt.n > 0) ? h->rt_dev : nullptr;
  a.sides = a.rt ? h->rs_dev : nullptr;
  a.n_wk = h->newton_explicit_kinetics ? 0 : h->n_wk;
  a.lane_mask = h->newton_mask;
  const int variant = newton_variant(h);
  a.ext_old = (h->nw_ext_old && !stationary) ? 1 : 0;
  a.bdf2 = (h->nw_bdf2_inline && !stationary) ? 1 : 0;
  a.bdf_hist0 = a.bdf2 ? h->nw_bdf_hist0 : 0;
  a.c_old2 = a.bdf2 ? h->c_old2 : nullptr;
  a.opt = &h->opt;
  const int64_t n_eff = newton_effective_batch(h);
  const bool use_lane4 = newton_lane4_preferred(N + 1, nx, n_eff, variant, h->opt);
  const bool use_lane2 = !use_lane4 && newton_lane2_preferred(N + 1, nx, n_eff, variant, h->opt);
  const bool use_lane = !use_lane4 && !use_lane2 && newton_lane_preferred(N + 1, nx, n_eff, variant, h->opt);
  if (use_lane4) {
    const size_t per_group = (newton_lane4_rec_doubles(N + 1, nx) + newton_lane4_state_doubles(N + 1, nx)) * sizeof(double);
    if (!h->lane4_buf) {
      int64_t groups = (h->cfg.batch_capacity + 7) / 8;
      const int64_t fit = (int64_t)(((size_t)48 << 30) / per_group);
      if (groups > fit) groups = fit;
      if (h->opt.newton_lane_groups >= 1 && h->opt.newton_lane_groups < groups) groups = h->opt.newton_lane_groups;
      if (groups < 1) groups = 1;
      HIP_TRY(h, dev_alloc(h, &h->lane4_buf, (size_t)groups * per_group / sizeof(double)));
      h->lane4_groups = groups;
    }
    const size_t vp2 = (size_t)((N + 2) / 2 * 2), cp2 = (size_t)((N + 1) / 2 * 2);
    a.lane_groups = h->lane4_groups;
    a.lane_ts = h->lane4_buf;
    a.lane_xs = a.lane_ts + (size_t)h->lane4_groups * vp2 * nx * 8;
    a.lane_tco = a.lane_xs + (size_t)h->lane4_groups * vp2 * nx * 8;
    a.lane_rec = a.lane_tco + (size_t)h->lane4_groups * cp2 * nx * 8;
    a.lane_tcn = a.lane_rec + (size_t)h->lane4_groups * newton_lane4_rec_doubles(N + 1, nx);
  } else if (use_lane2) {
    const size_t per_group = (newton_lane2_rec_doubles(N + 1, nx) + newton_lane2_state_doubles(N + 1, nx)) * sizeof(double);
    if (!h->lane2_buf) {
      int64_t groups = (h->cfg.batch_capacity + 15) / 16;
      const int64_t fit = (int64_t)(((size_t)48 << 30) / per_group);
      if (groups > fit) groups = fit;
      if (h->opt.newton_lane_groups >= 1 && h->opt.newton_lane_groups < groups) groups = h->opt.newton_lane_groups;
      if (groups < 1) groups = 1;
      HIP_TRY(h, dev_alloc(h, &h->lane2_buf, (size_t)groups * per_group / sizeof(double)));
      h->lane2_groups = groups;
    }
    const size_t vp2 = (size_t)((N + 2) / 2 * 2), cp2 = (size_t)((N + 1) / 2 * 2);
    a.lane_groups = h->lane2_groups;
    a.lane_ts = h->lane2_buf;
    a.lane_xs = a.lane_ts + (size_t)h->lane2_groups * vp2 * nx * 16;
    a.lane_tco = a.lane_xs + (size_t)h->lane2_groups * vp2 * nx * 16;
    a.lane_rec = a.lane_tco + (size_t)h->lane2_groups * cp2 * nx * 16;
    a.lane_tcn = a.lane_rec + (size_t)h->lane2_groups * newton_lane2_rec_doubles(N + 1, nx);
  } else if (use_lane) {
    // one operating point per lane: transposed state + records of as many groups of 32 operating points as the batch capacity has,
    // capped at 48 GiB (the launcher walks a larger batch in chunks)
    const size_t per_group = (newton_lane_rec_doubles(N + 1, nx) + newton_lane_state_doubles(N + 1, nx)) * sizeof(double);
    if (!h->lane_buf) {
      int64_t groups = (h->cfg.batch_capacity + 31) / 32;
      const int64_t fit = (int64_t)(((size_t)48 << 30) / per_group);
      if (groups > fit) groups = fit;
      if (h->opt.newton_lane_groups >= 1 && h->opt.newton_lane_groups < groups) groups = h->opt.newton_lane_groups;      // tests: several chunks
      if (groups < 1) groups = 1;
      HIP_TRY(h, dev_alloc(h, &h->lane_buf, (size_t)groups * per_group / sizeof(double)));
      h->lane_groups = groups;
    }
    a.lane_groups = h->lane_groups;
    const size_t vp2 = (size_t)((N + 2) / 2 * 2), cp2 = (size_t)((N + 1) / 2 * 2);
    a.lane_ts = h->lane_buf;
    a.lane_xs = a.lane_ts + (size_t)h->lane_groups * vp2 * nx * 32;
    a.lane_tco = a.lane_xs + (size_t)h->lane_groups * vp2 * nx * 32;
    a.lane_rec = a.lane_tco + (size_t)h->lane_groups * cp2 * nx * 32;
    a.lane_tcn = a.lane_rec + (size_t)h->lane_groups * newton_lane_rec_doubles(N + 1, nx);
  } else if (newton_sweep_preferred(N + 1, nx, h->B, a.rt ? 2 : (a.mpb ? 1 : 0), h->opt)) {
    // one team (N+1 lanes) per operating point, 64/(N+1) per wave; the workspace holds the records of the resident waves: at
    // most four per SIMD, all of the batch capacity, and 32 GiB
    const int64_t tpw = 64 / (N + 1);
    const size_t per_block = newton_sweep_doubles(N + 1, nx) * (size_t)tpw * sizeof(double);
    if (!h->sweep) {
      // (two teams per operating point: twice the waves for the same batch; the records of a wave are sized for tpw points either way)
      const int64_t per_wave = (N + 1 >= 6) ? tpw / 2 : tpw;
      int64_t blocks = (h->cfg.batch_capacity + per_wave - 1) / per_wave;
      if (blocks > 4096) blocks = 4096;
      if (h->opt.newton_sweep_blocks >= 1 && h->opt.newton_sweep_blocks < blocks) blocks = h->opt.newton_sweep_blocks;      // tests: several rounds per team
      const int64_t fit = (int64_t)(((size_t)32 << 30) / per_block);
      if (blocks > fit) blocks = fit;
      if (blocks < 1) blocks = 1;
      HIP_TRY(h, dev_alloc(h, &h->sweep, (size_t)blocks * per_block / sizeof(double)));
      h->sweep_blocks = (int)blocks;
    }
    a.sweep = h->sweep;
    a.sweep_stride = (int64_t)newton_sweep_doubles(N + 1, nx);
    a.sweep_blocks = h->sweep_blocks;
  }
  memcpy(a.wk_species, h->wk_species, sizeof(a.wk_species));
  memcpy(a.wk_nu, h->wk_nu, sizeof(a.wk_nu));
  memcpy(a.wk_alpha, h->wk_alpha, sizeof(a.wk_alpha));
  memcpy(a.wk_sat, h->wk_sat, sizeof(a.wk_sat));
  a.wk_k = h->wk_k;
  a.gw = h->gw;
  a.gv = h->gv;
  a.c = h->c;
  a.c_old = h->c_old;
  a.phi = h->v;
  a.pb = h->pb;
  a.flux = h->flux;
  a.cbulk = h->cbulk;
  a.status = h->status;
  a.iters = h->iters;
  int blocks = h->nw_blocks;
  if (h->opt.newton_blocks >= 1 && h->opt.newton_blocks < blocks) blocks = h->opt.newton_blocks;      // tuning: size of the persistent grid
  if ((int64_t)blocks > h->B) blocks = (int)h->B;
  if (use_lane4 || use_lane2 || use_lane) {
    const int rc = lane_order(h, a);
    if (rc != PNP_OK) return rc;
    if (a.B == 0) return PNP_OK;        // (a mask without a lane: nothing to solve)
  }
  if (use_lane4) HIP_TRY(h, launch_newton_lane4(a, h->stream));
  else if (use_lane2) HIP_TRY(h, launch_newton_lane2(a, h->stream));
  else if (use_lane) HIP_TRY(h, launch_newton_lane(a, h->stream));
  else HIP_TRY(h, launch_newton(a, blocks, h->stream));
  h->steps_done += nsteps;
  h->iters_valid = true;
  return PNP_OK;
}

int pnp_set_newton(pnp_handle* h, const pnp_newton_params* p, const double* mpb_radius) {
  if (!h || !p) return fail(h, PNP_EINVAL, "pnp_set_newton: null argument");
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_set_newton: the handle was not created with PNP_METHOD_NEWTON");
  if (p->struct_size != (int32_t)sizeof(pnp_newton_params)) return fail(h, PNP_EINVAL, "pnp_set_newton: struct_size mismatch (ABI)");
  if (p->wall_bc != 0 && p->wall_bc != 1) return fail(h, PNP_EINVAL, "pnp_set_newton: wall_bc must be 0 (Dirichlet) or 1 (Stern)");
  if (p->wall_bc == 1 && !(p->stern_capacitance > 0)) return fail(h, PNP_EINVAL, "pnp_set_newton: Stern capacitance must be positive");
  if (p->maxit < 1 || !(p->tol > 0)) return fail(h, PNP_EINVAL, "pnp_set_newton: maxit >= 1 and tol > 0 required");
  if (p->time_order < 0 || p->time_order > 2) return fail(h, PNP_EINVAL, "pnp_set_newton: time_order must be 0, 1 (backward Euler) or 2 (BDF2)");
  if (p->time_order != h->np.time_order || p->predictor != h->np.predictor) h->bdf_history = false;
  if (p->predictor != 0 && p->predictor != 1) return fail(h, PNP_EINVAL, "pnp_set_newton: predictor must be 0 or 1");
  h->np = *p;
  h->mpb = false;
  for (int k = 0; k < h->a.N; ++k) {
    const double a = mpb_radius ? mpb_radius[k] : 0.0;
    if (!(a >= 0)) return fail(h, PNP_EINVAL, "pnp_set_newton: negative MPB radius");
    h->volk[k] = 6.022140857e23 * a * a * a;       // unit_NA, catint/units.py
    if (h->volk[k] != 0.0) h->mpb = true;
  }
  return PNP_OK;
}

int pnp_set_convection(pnp_handle* h, double velocity) {
  if (!h) return PNP_EINVAL;
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_set_convection: physical mode only");
  if (!(std::fabs(velocity) < 1e300)) return fail(h, PNP_EINVAL, "pnp_set_convection: velocity must be finite");
  h->velocity = velocity;
  return PNP_OK;
}

int pnp_set_grid(pnp_handle* h, const double* x) {
  if (!h || !x) return fail(h, PNP_EINVAL, "pnp_set_grid: null argument");
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_set_grid: only the physical mode takes a non-uniform grid");
  const int nx = h->a.nx;
  const double dx = h->a.dx;
  std::vector<double> w((size_t)nx, 1.0), v((size_t)nx, 0.0);
  for (int i = 0; i + 1 < nx; ++i) {
    const double hh = x[i + 1] - x[i];
    if (!(hh > 0)) return fail(h, PNP_EINVAL, "pnp_set_grid: x must be strictly increasing");
    w[i] = dx / hh;
    v[i] += 0.5 * hh / dx;
    v[i + 1] += 0.5 * hh / dx;
  }
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(h->gw, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->gv, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->xgrid.assign(x, x + nx);
  return PNP_OK;
}

int pnp_set_wall_kinetics(pnp_handle* h, int32_t n, const int32_t* species, const double* nu, const double* k) {
  if (!h) return PNP_EINVAL;
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_set_wall_kinetics: the handle was not created with PNP_METHOD_NEWTON");
  if (n < 0 || n > PNP_MAX_WALL_REACTIONS) return fail(h, PNP_EINVAL, "pnp_set_wall_kinetics: at most 8 surface reactions");
  if (n == 0) {
    h->n_wk = 0;
    return PNP_OK;
  }
  if (!species || !nu || !k) return fail(h, PNP_EINVAL, "pnp_set_wall_kinetics: null argument");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_set_wall_kinetics: call pnp_set_batch first (rate constants are per lane)");
  const int N = h->a.N;
  for (int r = 0; r < n; ++r) {
    if (species[r] < -1 || species[r] >= N) return fail(h, PNP_EINVAL, "pnp_set_wall_kinetics: species index out of range");
    h->wk_species[r] = species[r];
    h->wk_alpha[r] = h->wk_sat[r] = 0.0;      // a new table is first order until pnp_set_wall_rate_law says otherwise
    for (int kk = 0; kk < PNP_NEWTON_MAX_SPECIES; ++kk) h->wk_nu[r][kk] = kk < N ? nu[r * N + kk] : 0.0;
  }
  std::vector<double> kp((size_t)h->B * PNP_MAX_WALL_REACTIONS, 0.0);
  for (int64_t b = 0; b < h->B; ++b)
    for (int r = 0; r < n; ++r) kp[(size_t)b * PNP_MAX_WALL_REACTIONS + r] = k[b * n + r];
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(h->wk_k, kp.data(), kp.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->n_wk = n;
  return PNP_OK;
}

int pnp_set_wall_rate_law(pnp_handle* h, int32_t n, const double* alpha, const double* saturation) {
  if (!h) return PNP_EINVAL;
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_set_wall_rate_law: the handle was not created with PNP_METHOD_NEWTON");
  if (n != h->n_wk) return fail(h, PNP_ESTATE, "pnp_set_wall_rate_law: n differs from the table of pnp_set_wall_kinetics (call that first)");
  for (int r = 0; r < n; ++r) {
    const double al = alpha ? alpha[r] : 0.0, ks = saturation ? saturation[r] : 0.0;
    if (!(al == al) || !(ks >= 0.0)) return fail(h, PNP_EINVAL, "pnp_set_wall_rate_law: alpha must be a number, saturation >= 0");
    if (ks != 0.0 && h->wk_species[r] < 0) return fail(h, PNP_EINVAL, "pnp_set_wall_rate_law: saturation on a zeroth-order reaction");
  }
  for (int r = 0; r < n; ++r) {
    h->wk_alpha[r] = alpha ? alpha[r] : 0.0;
    h->wk_sat[r] = saturation ? saturation[r] : 0.0;
  }
  return PNP_OK;
}

int pnp_solve_stationary(pnp_handle* h, double tol, int32_t maxit, int32_t* status) {
  if (!h) return PNP_EINVAL;
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_solve_stationary: the handle was not created with PNP_METHOD_NEWTON");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_solve_stationary: call pnp_set_batch first");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const int rc = run_newton(h, 1, true, tol, maxit);
  if (rc != PNP_OK) return rc;
  if (status) return pnp_get_status(h, status);
  return PNP_OK;
}

int pnp_solve_surface(pnp_handle* h, const double* flux, int32_t nsteps, double* csurf, double* vsurf, double* esurf,
                      int32_t* status) {
  if (!h) return PNP_EINVAL;
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_solve_surface: the handle was not created with PNP_METHOD_NEWTON");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_solve_surface: call pnp_set_batch first");
  if (nsteps < 0) return fail(h, PNP_EINVAL, "pnp_solve_surface: nsteps < 0");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const int64_t B = h->B;
  const int N = h->a.N, ldx = h->a.ldx;
  // everything below is queued on the handle's stream; one synchronisation at the end
  if (flux) HIP_TRY(h, hipMemcpyAsync(h->flux, flux, (size_t)B * N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  const int rc = nsteps == 0 ? run_newton(h, 1, true, 0.0, 0) : newton_timesteps(h, nsteps);
  if (rc != PNP_OK) return rc;
  std::vector<double> p01;
  if (csurf) {
    HIP_TRY(h, launch_surface(h->a, h->csurf, h->stream));
    HIP_TRY(h, hipMemcpyAsync(csurf, h->csurf, (size_t)B * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  if (vsurf || esurf) {
    p01.resize((size_t)B * 2);
    HIP_TRY(h, hipMemcpy2DAsync(p01.data(), 2 * sizeof(double), h->v, (size_t)ldx * sizeof(double), 2 * sizeof(double), (size_t)B,
                                hipMemcpyDeviceToHost, h->stream));
  }
  if (status) HIP_TRY(h, hipMemcpyAsync(status, h->status, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int64_t b = 0; b < B; ++b) {
    if (vsurf) vsurf[b] = p01[2 * b];
    if (esurf) esurf[b] = -(p01[2 * b + 1] - p01[2 * b]) / (h->xgrid[1] - h->xgrid[0]);
  }
  return PNP_OK;
}

int pnp_scf_cycle(pnp_handle* h, const pnp_scf_params* p, const double* nel, const double* nprod, pnp_scf_state* s,
                  int32_t* iterations) {
  if (!h || !p || !s) return fail(h, PNP_EINVAL, "pnp_scf_cycle: null argument");
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_scf_cycle: the handle was not created with PNP_METHOD_NEWTON");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_scf_cycle: call pnp_set_batch first");
  if (p->struct_size != (int32_t)sizeof(pnp_scf_params)) return fail(h, PNP_EINVAL, "pnp_scf_cycle: struct_size mismatch (ABI)");
  if (h->n_wk < 1) return fail(h, PNP_ESTATE, "pnp_scf_cycle: no kinetic model (pnp_set_wall_kinetics)");
  if (p->istep < 1) return fail(h, PNP_EINVAL, "pnp_scf_cycle: the first iteration (transport solve from the bulk state) is the caller's");
  const int N = h->a.N;
  if (p->species_H >= N || p->species_OH >= N) return fail(h, PNP_EINVAL, "pnp_scf_cycle: pH species index out of range");
  if (!s->surface_concentration || !s->surface_concentration_old || !s->flux || !s->current_density_old || !s->mix ||
      !s->accuracy || !s->surface_pH || !s->surface_potential || !s->surface_efield || !s->step_to_check || !s->active || !s->failed)
    return fail(h, PNP_EINVAL, "pnp_scf_cycle: null array in pnp_scf_state");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const int64_t B = h->B, cap = h->cfg.batch_capacity;
  if (!h->scf_d) {
    HIP_TRY(h, dev_alloc(h, &h->scf_d, (size_t)cap * (4 * N + 5)));
    HIP_TRY(h, dev_alloc(h, &h->scf_i, (size_t)cap * 3 + 65));
    HIP_TRY(h, dev_alloc(h, &h->scf_snap, (size_t)cap * (N + 1) * h->a.ldx));
  }
  ScfArgs a;
  memset(&a, 0, sizeof(a));
  a.N = N;
  a.n_wk = h->n_wk;
  a.iH = p->species_H;
  a.iOH = p->species_OH;
  a.ldx = h->a.ldx;
  a.B = B;
  a.tau = p->tau_scf;
  a.h0 = h->xgrid[1] - h->xgrid[0];
  a.faraday = p->faraday;
  memcpy(a.wk_species, h->wk_species, sizeof(a.wk_species));
  memcpy(a.wk_nu, h->wk_nu, sizeof(a.wk_nu));
  memcpy(a.wk_alpha, h->wk_alpha, sizeof(a.wk_alpha));
  memcpy(a.wk_sat, h->wk_sat, sizeof(a.wk_sat));
  a.pb = h->pb;
  for (int k = 0; k < N; ++k) {
    a.nel[k] = nel ? nel[k] : 1.0;
    a.nprod[k] = nprod ? nprod[k] : 1.0;
  }
  a.wk_k = h->wk_k;
  double* d = h->scf_d;
  a.sc = d;
  a.sc_old = d + (size_t)cap * N;
  a.flux = h->flux;                       // the transport solve reads its wall fluxes here
  a.cd_old = d + (size_t)cap * 2 * N;
  double* scal = d + (size_t)cap * 3 * N;  // (one [B][N] block spare)
  scal += (size_t)cap * N;
  a.mix = scal;
  a.acc = scal + cap;
  a.surface_pH = scal + 2 * cap;
  a.vsurf = scal + 3 * cap;
  a.esurf = scal + 4 * cap;
  a.step_to_check = h->scf_i;
  a.active = h->scf_i + cap;
  a.failed = h->scf_i + 2 * cap;
  a.counters = h->scf_i + 3 * cap;
  a.c = h->c;
  a.phi = h->v;
  a.snap_c = h->scf_snap;
  a.snap_phi = h->scf_snap + (size_t)cap * N * h->a.ldx;
  a.status = h->status;
  const size_t bn = (size_t)B * N * sizeof(double), bd = (size_t)B * sizeof(double), bi = (size_t)B * sizeof(int32_t);
  hipStream_t st = h->stream;
  HIP_TRY(h, hipMemcpyAsync(a.sc, s->surface_concentration, bn, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(a.sc_old, s->surface_concentration_old, bn, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(a.flux, s->flux, bn, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(a.cd_old, s->current_density_old, bn, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(a.mix, s->mix, bd, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(a.acc, s->accuracy, bd, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(a.surface_pH, s->surface_pH, bd, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(a.vsurf, s->surface_potential, bd, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(a.esurf, s->surface_efield, bd, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(a.step_to_check, s->step_to_check, bi, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(a.active, s->active, bi, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(a.failed, s->failed, bi, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemsetAsync(a.counters, 0, 65 * sizeof(int32_t), st));
  HIP_TRY(h, hipMemcpyAsync(a.snap_c, a.c, (size_t)B * N * h->a.ldx * sizeof(double), hipMemcpyDeviceToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(a.snap_phi, a.phi, (size_t)B * h->a.ldx * sizeof(double), hipMemcpyDeviceToDevice, st));
  const int every = p->check_every > 0 ? (p->check_every < 32 ? p->check_every : 32) : 8;
  int rc = PNP_OK;
  int32_t left = 1;
  h->newton_mask = a.active;
  h->newton_explicit_kinetics = true;
  for (int istep = p->istep + 1; istep <= p->max_iter && rc == PNP_OK; ++istep) {
    a.istep = istep;
    a.slot = istep & 63;
    hipError_t e = hipMemsetAsync(a.counters + a.slot, 0, sizeof(int32_t), st);
    if (e == hipSuccess) e = launch_scf_pre(a, st);
    if (e != hipSuccess) {
      rc = fail(h, PNP_EDEVICE, std::string("pnp_scf_cycle: ") + hipGetErrorString(e));
      break;
    }
    rc = run_newton(h, 1, true, 0.0, 0);       // stationary, warm: the state of the previous iteration (restart=True, calculator.py:523)
    if (rc != PNP_OK) break;
    e = launch_scf_keep(a, st);
    if (e == hipSuccess) e = launch_scf_post(a, st);
    if (e == hipSuccess && ((istep - p->istep) % every == 0 || istep == p->max_iter)) {
      e = hipMemcpyAsync(&left, a.counters + a.slot, sizeof(int32_t), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e == hipSuccess && left == 0) break;
    }
    if (e != hipSuccess) rc = fail(h, PNP_EDEVICE, std::string("pnp_scf_cycle: ") + hipGetErrorString(e));
  }
  h->newton_mask = nullptr;
  h->newton_explicit_kinetics = false;
  if (rc != PNP_OK) return rc;
  int32_t last = 0;
  HIP_TRY(h, hipMemcpyAsync(s->surface_concentration, a.sc, bn, hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemcpyAsync(s->surface_concentration_old, a.sc_old, bn, hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemcpyAsync(s->flux, a.flux, bn, hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemcpyAsync(s->current_density_old, a.cd_old, bn, hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemcpyAsync(s->mix, a.mix, bd, hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemcpyAsync(s->accuracy, a.acc, bd, hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemcpyAsync(s->surface_pH, a.surface_pH, bd, hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemcpyAsync(s->surface_potential, a.vsurf, bd, hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemcpyAsync(s->surface_efield, a.esurf, bd, hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemcpyAsync(s->step_to_check, a.step_to_check, bi, hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemcpyAsync(s->active, a.active, bi, hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemcpyAsync(s->failed, a.failed, bi, hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemcpyAsync(&last, a.counters + 64, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipStreamSynchronize(st));
  if (iterations) *iterations = last > p->istep ? last : p->istep;
  return PNP_OK;
}

int pnp_get_newton_iterations(pnp_handle* h, int32_t* iters) {
  if (!h || !iters) return fail(h, PNP_EINVAL, "pnp_get_newton_iterations: null argument");
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_get_newton_iterations: the handle was not created with PNP_METHOD_NEWTON");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_get_newton_iterations: call pnp_set_batch first");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(iters, h->iters, (size_t)h->B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return PNP_OK;
}

int pnp_set_potential(pnp_handle* h, const double* phi) {
  if (!h || !phi) return fail(h, PNP_EINVAL, "pnp_set_potential: null argument");
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_set_potential: the handle was not created with PNP_METHOD_NEWTON");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_set_potential: call pnp_set_batch first");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const size_t w = (size_t)h->a.nx * sizeof(double), dp = (size_t)h->a.ldx * sizeof(double);
  HIP_TRY(h, hipMemcpy2DAsync(h->v, dp, phi, w, w, (size_t)h->B, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return PNP_OK;
}

int pnp_set_lanes(pnp_handle* h, int64_t n, const int64_t* lanes, const double* c, const double* phi) {
  if (!h || (n > 0 && (!lanes || !c))) return fail(h, PNP_EINVAL, "pnp_set_lanes: null argument");
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_set_lanes: the handle was not created with PNP_METHOD_NEWTON");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_set_lanes: call pnp_set_batch first");
  if (n < 0) return fail(h, PNP_EINVAL, "pnp_set_lanes: n < 0");
  for (int64_t i = 0; i < n; ++i)
    if (lanes[i] < 0 || lanes[i] >= h->B) return fail(h, PNP_EINVAL, "pnp_set_lanes: lane index out of range");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  h->bdf_history = false;        // (the patched lanes have no previous time level of their own: the next BDF2 step starts over)
  const int N = h->a.N, nx = h->a.nx, ldx = h->a.ldx;
  const size_t w = (size_t)nx * sizeof(double), dp = (size_t)ldx * sizeof(double);
  // a handful of lanes: one strided copy per lane (N rows of the concentrations, one of the potential), all on the handle's stream
  for (int64_t i = 0; i < n; ++i) {
    HIP_TRY(h, hipMemcpy2DAsync(h->c + (size_t)lanes[i] * N * ldx, dp, c + (size_t)i * N * nx, w, w, (size_t)N, hipMemcpyHostToDevice,
                                h->stream));
    if (phi) HIP_TRY(h, hipMemcpyAsync(h->v + (size_t)lanes[i] * ldx, phi + (size_t)i * nx, w, hipMemcpyHostToDevice, h->stream));
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return PNP_OK;
}

int pnp_set_lane_mask(pnp_handle* h, const int32_t* mask) {
  if (!h) return PNP_EINVAL;
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_set_lane_mask: the handle was not created with PNP_METHOD_NEWTON");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_set_lane_mask: call pnp_set_batch first");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (!mask) {
    h->newton_mask = nullptr;
    return PNP_OK;
  }
  h->user_mask_host.assign(mask, mask + h->B);
  h->user_mask_count = 0;
  for (int64_t b = 0; b < h->B; ++b) h->user_mask_count += mask[b] != 0 ? 1 : 0;
  if (!h->user_mask) HIP_TRY(h, dev_alloc(h, &h->user_mask, (size_t)h->cfg.batch_capacity));
  HIP_TRY(h, hipMemcpyAsync(h->user_mask, mask, (size_t)h->B * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->newton_mask = h->user_mask;
  return PNP_OK;
}

// ---- pnp_autotune: the kernel family of the physical mode chosen by measurement on THIS device and THIS batch --------------------
// (the selection functions carry thresholds measured on the devices of one pool, and devices differ by ~10 %: DESIGN.md section 6a)
static const struct {
  const char* name;
  int kernel, fused;
} kTuneChoices[PNP_AUTOTUNE_CHOICES] = {{"lane4", NK_LANE4, -1},         {"lane2", NK_LANE2, -1}, {"lane", NK_LANE, 0},    {"lane+fused", NK_LANE, 1},
                                        {"workgroup", NK_WORKGROUP, -1}, {"team", NK_TEAM, -1},   {"sweep", NK_SWEEP, -1}, {"both", NK_BOTH, -1}};

const char* pnp_autotune_name(int32_t i) { return (i >= 0 && i < PNP_AUTOTUNE_CHOICES) ? kTuneChoices[i].name : nullptr; }

static void free_dev(pnp_handle* h, double** p, size_t bytes) {
  if (*p) {
    (void)hipFree(*p);
    h->dev_bytes -= (int64_t)bytes;
  }
  *p = nullptr;
}

int32_t pnp_autotune_default(const pnp_handle* h) {
  if (!h || !h->newton || !h->have_batch) return -1;
  Options o = h->opt;
  o.newton_kernel = NK_AUTO;
  o.lane_fused = -1;
  const int nb = h->a.N + 1, nx = h->a.nx, variant = newton_variant(h);
  const int64_t n_eff = newton_effective_batch(h);
  if (newton_lane4_preferred(nb, nx, n_eff, variant, o)) return 0;
  if (newton_lane2_preferred(nb, nx, n_eff, variant, o)) return 1;
  if (newton_lane_preferred(nb, nx, n_eff, variant, o)) return 3;      // (timesteps: fused at every batch, pnp_lane.hip: launch_lane_nb)
  return 4;
}

int pnp_autotune(pnp_handle* h, int32_t nsteps, double* ms_per_step, int32_t* chosen) {
  if (!h) return PNP_EINVAL;
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_autotune: the handle was not created with PNP_METHOD_NEWTON");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_autotune: call pnp_set_batch first");
  if (nsteps < 1 || nsteps > 64) return fail(h, PNP_EINVAL, "pnp_autotune: nsteps must be 1 ... 64");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const int N = h->a.N, nx = h->a.nx, ldx = h->a.ldx, nb = N + 1;
  const int64_t B = h->B;
  const int variant = newton_variant(h);
  const size_t nc = (size_t)B * N * ldx, nv = (size_t)B * ldx;
  // the trial steps run on the handle's own state: it is put back after every trial, with everything a step moves along
  double *save_c = nullptr, *save_v = nullptr, *save_c2 = nullptr, *save_p2 = nullptr;
  int32_t *save_st = nullptr, *save_it = nullptr;
  struct Cleanup {
    pnp_handle* h;
    double **a, **b, **c, **d;
    int32_t **e, **f;
    ~Cleanup() {
      for (double** p : {a, b, c, d})
        if (*p) (void)hipFree(*p);
      for (int32_t** p : {e, f})
        if (*p) (void)hipFree(*p);
    }
  } cleanup{h, &save_c, &save_v, &save_c2, &save_p2, &save_st, &save_it};
  HIP_TRY(h, hipMalloc((void**)&save_c, nc * sizeof(double)));
  HIP_TRY(h, hipMalloc((void**)&save_v, nv * sizeof(double)));
  HIP_TRY(h, hipMalloc((void**)&save_st, (size_t)B * sizeof(int32_t)));
  HIP_TRY(h, hipMalloc((void**)&save_it, (size_t)B * sizeof(int32_t)));
  if (h->c_old2) HIP_TRY(h, hipMalloc((void**)&save_c2, nc * sizeof(double)));
  if (h->phi_old2) HIP_TRY(h, hipMalloc((void**)&save_p2, nv * sizeof(double)));
  auto copy = [&](bool back) -> hipError_t {
    struct {
      void *live, *kept;
      size_t bytes;
    } items[6] = {{h->c, save_c, nc * sizeof(double)},        {h->v, save_v, nv * sizeof(double)},
                  {h->c_old2, save_c2, nc * sizeof(double)},  {h->phi_old2, save_p2, nv * sizeof(double)},
                  {h->status, save_st, (size_t)B * sizeof(int32_t)}, {h->iters, save_it, (size_t)B * sizeof(int32_t)}};
    for (auto& it : items) {
      if (!it.live || !it.kept) continue;
      const hipError_t e = hipMemcpyAsync(back ? it.live : it.kept, back ? it.kept : it.live, it.bytes, hipMemcpyDeviceToDevice, h->stream);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  };
  HIP_TRY(h, copy(false));
  const Options opt0 = h->opt;
  const int64_t steps0 = h->steps_done;
  const bool hist0 = h->bdf_history, iters0 = h->iters_valid;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIP_TRY(h, hipEventCreate(&e0));
  HIP_TRY(h, hipEventCreate(&e1));
  double best = -1.0;
  int best_i = -1, rc = PNP_OK;
  int64_t best_ok = -1;
  std::vector<int32_t> st((size_t)B);
  // the device's clocks settle over ~0.1 s of load: without this the first families tried are timed on a cold device (measured: the
  // lane-quad kernel, first in the list, 11.7 ms per step against 7.6 ms warm at 8192 x 8 x 512)
  for (auto t0 = std::chrono::steady_clock::now(); rc == PNP_OK && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(200);) {
    h->steps_done = steps0;
    h->bdf_history = hist0;
    h->iters_valid = iters0;
    if (copy(true) != hipSuccess) rc = fail(h, PNP_EDEVICE, "pnp_autotune: restoring the state failed");
    if (rc == PNP_OK) rc = newton_timesteps(h, 1);
    if (rc == PNP_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, PNP_EDEVICE, "pnp_autotune: synchronisation failed");
  }
  for (int i = 0; i < PNP_AUTOTUNE_CHOICES && rc == PNP_OK; ++i) {
    if (ms_per_step) ms_per_step[i] = -1.0;
    const int k = kTuneChoices[i].kernel;
    const bool applicable = k == NK_LANE4   ? newton_lane4_supported(nb, nx, variant)
                            : k == NK_LANE2 ? newton_lane2_supported(nb, nx, variant)
                            : k == NK_LANE  ? newton_lane_supported(nb, nx, variant)
                            : k == NK_TEAM  ? (nb >= 3 && h->work != nullptr)
                            : k == NK_SWEEP ? nb >= 3
                            : k == NK_BOTH  ? (nb >= 6 && nx >= 8)
                                            : true;
    if (!applicable) continue;
    h->opt = opt0;
    h->opt.newton_kernel = k;
    h->opt.lane_fused = kTuneChoices[i].fused;
    float ms = 0.0f;
    for (int pass = 0; pass < 2 && rc == PNP_OK; ++pass) {      // the first pass allocates the family's workspace and warms the caches
      h->steps_done = steps0;
      h->bdf_history = hist0;
      h->iters_valid = iters0;
      if (copy(true) != hipSuccess) rc = fail(h, PNP_EDEVICE, "pnp_autotune: restoring the state failed");
      if (rc == PNP_OK && hipEventRecord(e0, h->stream) != hipSuccess) rc = fail(h, PNP_EDEVICE, "pnp_autotune: hipEventRecord");
      if (rc == PNP_OK) rc = newton_timesteps(h, pass == 0 ? 1 : nsteps);
      if (rc == PNP_OK && (hipEventRecord(e1, h->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                           hipEventElapsedTime(&ms, e0, e1) != hipSuccess))
        rc = fail(h, PNP_EDEVICE, "pnp_autotune: timing a trial failed");
    }
    if (rc != PNP_OK) break;
    if (hipMemcpy(st.data(), h->status, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) {
      rc = fail(h, PNP_EDEVICE, "pnp_autotune: reading the status failed");
      break;
    }
    int64_t ok = 0;
    for (int64_t b = 0; b < B; ++b) ok += st[(size_t)b] == PNP_STATUS_OK ? 1 : 0;
    const double per_step = (double)ms / nsteps;
    if (ms_per_step) ms_per_step[i] = per_step;
    // the fastest family among those that solved the most operating points
    if (ok > best_ok || (ok == best_ok && per_step < best)) {
      best = per_step;
      best_i = i;
      best_ok = ok;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  h->opt = opt0;
  h->steps_done = steps0;
  h->bdf_history = hist0;
  h->iters_valid = iters0;
  if (copy(true) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return fail(h, PNP_EDEVICE, "pnp_autotune: restoring the state failed");
  if (rc != PNP_OK) return rc;
  if (best_i < 0) return fail(h, PNP_ESTATE, "pnp_autotune: no kernel family applies");
  h->opt.newton_kernel = kTuneChoices[best_i].kernel;
  h->opt.lane_fused = kTuneChoices[best_i].fused;
  // the workspaces of the families that lost go back to the device
  const int nk = h->opt.newton_kernel;
  if (nk != NK_LANE) free_dev(h, &h->lane_buf, (size_t)h->lane_groups * (newton_lane_rec_doubles(nb, nx) + newton_lane_state_doubles(nb, nx)) * sizeof(double));
  if (nk != NK_LANE2)
    free_dev(h, &h->lane2_buf, (size_t)h->lane2_groups * (newton_lane2_rec_doubles(nb, nx) + newton_lane2_state_doubles(nb, nx)) * sizeof(double));
  if (nk != NK_LANE4)
    free_dev(h, &h->lane4_buf, (size_t)h->lane4_groups * (newton_lane4_rec_doubles(nb, nx) + newton_lane4_state_doubles(nb, nx)) * sizeof(double));
  if (nk == NK_LANE || nk == NK_LANE2 || nk == NK_LANE4 || nk == NK_TEAM)
    free_dev(h, &h->sweep, (size_t)h->sweep_blocks * newton_sweep_doubles(nb, nx) * (size_t)(64 / nb) * sizeof(double));
  if (chosen) *chosen = best_i;
  return PNP_OK;
}

// ---- pnp_tune_placement: the lane kernels' workspace put where it runs fastest ----------------------------------------------------------
// The rate of an HBM-bound lane-kernel launch depends on WHERE its workspace lies in device memory: on some devices one of three
// discrete values (1.52 / 1.66 / 1.83e6 timesteps/s at 32 768 x 8 x 512), fixed for the lifetime of the allocation, equal between
// repetitions to 0.3 %, different between allocations of one process (tools/probe/lane_modes2.py, profiles/r04_lane_modes.jsonl;
// address translation, instruction cache, start phases and the layout of the streams ruled out: profiles/r04_measured_not_taken.txt #11).
// So: allocate the workspace up to `trials` times (the earlier ones stay allocated meanwhile, so that each lands elsewhere), time
// `nsteps` timesteps on each from the same state, keep the fastest, free the others.
int pnp_tune_placement(pnp_handle* h, int32_t nsteps, int32_t trials, double* ms_per_step) {
  if (!h) return PNP_EINVAL;
  if (!h->newton) return fail(h, PNP_EINVAL, "pnp_tune_placement: the handle was not created with PNP_METHOD_NEWTON");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_tune_placement: call pnp_set_batch first");
  if (nsteps < 1 || nsteps > 64 || trials < 1 || trials > 16) return fail(h, PNP_EINVAL, "pnp_tune_placement: nsteps 1 ... 64, trials 1 ... 16");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  for (int i = 0; i < trials && ms_per_step; ++i) ms_per_step[i] = -1.0;
  const int N = h->a.N, nx = h->a.nx, ldx = h->a.ldx, nb = N + 1, variant = newton_variant(h);
  const int64_t B = h->B, n_eff = newton_effective_batch(h);
  double** bufp = nullptr;
  int64_t* groupsp = nullptr;
  size_t per_group = 0;
  if (newton_lane4_preferred(nb, nx, n_eff, variant, h->opt)) {
    bufp = &h->lane4_buf, groupsp = &h->lane4_groups, per_group = newton_lane4_rec_doubles(nb, nx) + newton_lane4_state_doubles(nb, nx);
  } else if (newton_lane2_preferred(nb, nx, n_eff, variant, h->opt)) {
    bufp = &h->lane2_buf, groupsp = &h->lane2_groups, per_group = newton_lane2_rec_doubles(nb, nx) + newton_lane2_state_doubles(nb, nx);
  } else if (newton_lane_preferred(nb, nx, n_eff, variant, h->opt)) {
    bufp = &h->lane_buf, groupsp = &h->lane_groups, per_group = newton_lane_rec_doubles(nb, nx) + newton_lane_state_doubles(nb, nx);
  }
  if (!bufp) return PNP_OK;      // (no lane kernel for this batch: nothing to place)
  const size_t nc = (size_t)B * N * ldx, nv = (size_t)B * ldx;
  struct Saved {
    void *live, *kept;
    size_t bytes;
  } items[6] = {{h->c, nullptr, nc * sizeof(double)},        {h->v, nullptr, nv * sizeof(double)},
                {h->c_old2, nullptr, nc * sizeof(double)},  {h->phi_old2, nullptr, nv * sizeof(double)},
                {h->status, nullptr, (size_t)B * sizeof(int32_t)}, {h->iters, nullptr, (size_t)B * sizeof(int32_t)}};
  std::vector<double*> held;      // the workspaces of the trials so far
  auto release = [&]() {      // the snapshot, and every workspace but the one the handle keeps
    for (auto& it : items)
      if (it.kept) (void)hipFree(it.kept);
    for (double* p : held)
      if (p && p != *bufp) {
        (void)hipFree(p);
        h->dev_bytes -= (int64_t)((size_t)*groupsp * per_group * sizeof(double));
      }
    held.clear();
  };
  for (auto& it : items)
    if (it.live && hipMalloc(&it.kept, it.bytes) != hipSuccess) {
      it.kept = nullptr;
      release();
      return fail(h, PNP_ENOMEM, "pnp_tune_placement: no memory for the state snapshot");
    }
  auto copy = [&](bool back) -> bool {
    for (auto& it : items)
      if (it.live && it.kept &&
          hipMemcpyAsync(back ? it.live : it.kept, back ? it.kept : it.live, it.bytes, hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
        return false;
    return true;
  };
  const int64_t steps0 = h->steps_done;
  const bool hist0 = h->bdf_history, iters0 = h->iters_valid;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = PNP_OK, best_i = -1;
  double best = 0.0;
  double* best_buf = *bufp;
  if (!copy(false) || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = fail(h, PNP_EDEVICE, "pnp_tune_placement: set-up failed");
  // (clocks up before the first placement is timed: ~0.2 s of the same launches, see pnp_autotune)
  for (auto t0 = std::chrono::steady_clock::now(); rc == PNP_OK && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(200);) {
    h->steps_done = steps0;
    h->bdf_history = hist0;
    h->iters_valid = iters0;
    if (!copy(true)) rc = fail(h, PNP_EDEVICE, "pnp_tune_placement: restoring the state failed");
    if (rc == PNP_OK) rc = newton_timesteps(h, 1);
    if (rc == PNP_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, PNP_EDEVICE, "pnp_tune_placement: synchronisation failed");
  }
  for (int i = 0; i < trials && rc == PNP_OK; ++i) {
    if (i > 0) {
      // the next placement: run_newton allocates on first use, while the earlier workspaces are still there -- if the device has room for
      // it twice over (other handles and other processes on the device allocate too), else what was measured so far decides
      size_t free_b = 0, total_b = 0;
      const size_t W = (size_t)*groupsp * per_group * sizeof(double);
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 2 * W + ((size_t)4 << 30)) break;
      *bufp = nullptr;
    }
    float ms = 0.0f;
    for (int pass = 0; pass < 2 && rc == PNP_OK; ++pass) {      // (the first pass allocates and warms)
      h->steps_done = steps0;
      h->bdf_history = hist0;
      h->iters_valid = iters0;
      if (!copy(true)) rc = fail(h, PNP_EDEVICE, "pnp_tune_placement: restoring the state failed");
      if (rc == PNP_OK && hipEventRecord(e0, h->stream) != hipSuccess) rc = fail(h, PNP_EDEVICE, "pnp_tune_placement: hipEventRecord");
      if (rc == PNP_OK) rc = newton_timesteps(h, pass == 0 ? 1 : nsteps);
      if (rc == PNP_OK && (hipEventRecord(e1, h->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                           hipEventElapsedTime(&ms, e0, e1) != hipSuccess))
        rc = fail(h, PNP_EDEVICE, "pnp_tune_placement: timing a trial failed");
    }
    if (rc == PNP_ENOMEM && i > 0) {      // no room for another placement: what was measured so far decides
      rc = PNP_OK;
      if (*bufp) held.push_back(*bufp);
      break;
    }
    if (rc != PNP_OK) break;
    held.push_back(*bufp);
    const double per_step = (double)ms / nsteps;
    if (ms_per_step) ms_per_step[i] = per_step;
    if (best_i < 0 || per_step < best) {
      best = per_step;
      best_i = i;
      best_buf = *bufp;
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (*bufp && std::find(held.begin(), held.end(), *bufp) == held.end()) held.push_back(*bufp);      // (a trial that failed after allocating)
  *bufp = best_buf;
  h->steps_done = steps0;
  h->bdf_history = hist0;
  h->iters_valid = iters0;
  const bool restored = copy(true) && hipStreamSynchronize(h->stream) == hipSuccess;
  release();
  if (!restored) return fail(h, PNP_EDEVICE, "pnp_tune_placement: restoring the state failed");
  return rc;
}

int pnp_step(pnp_handle* h, int32_t nsteps, int32_t steps_per_launch) {
  if (!h) return PNP_EINVAL;
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_step: call pnp_set_batch first");
  if (nsteps < 0) return fail(h, PNP_EINVAL, "pnp_step: nsteps < 0");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (h->newton) {
    if (nsteps == 0) return PNP_OK;
    return newton_timesteps(h, nsteps);
  }
  int spl = steps_per_launch <= 0 ? 256 : steps_per_launch;   // kernel boundaries cost ~6 us each
  if (h->a.has_rates) spl = 1;
  const int S = step_streams(h, (nsteps + spl - 1) / spl);
  if (S > 1) {
    // fork: the chunk streams start behind everything queued on the handle's stream; join: the handle's stream continues behind
    // the last launch of every chunk.  In between each chunk's launches depend only on that chunk's previous launch.
    if (!h->ev_fork) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < S - 1; ++i) {
      if (!h->aux_stream[i]) HIP_TRY(h, hipStreamCreateWithFlags(&h->aux_stream[i], hipStreamNonBlocking));
      if (!h->ev_join[i]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
    }
    HIP_TRY(h, hipEventRecord(h->ev_fork, h->stream));
    for (int i = 0; i < S - 1; ++i) HIP_TRY(h, hipStreamWaitEvent(h->aux_stream[i], h->ev_fork, 0));
  }
  const int64_t B = h->a.B;
  int left = nsteps;
  int rc = PNP_OK;
  while (left > 0 && rc == PNP_OK) {
    const int n = left < spl ? left : spl;
    if (S == 1) {
      rc = run_steps(h, n);
    } else {
      for (int i = 0; i < S && rc == PNP_OK; ++i) {
        const int64_t b0 = B * i / S, b1 = B * (i + 1) / S;
        rc = run_steps_on(h, n, b0, b1 - b0, i == 0 ? h->stream : h->aux_stream[i - 1]);
      }
      if (rc == PNP_OK) {
        if (n & 1) h->cur = 1 - h->cur;
        h->steps_done += n;
      }
    }
    left -= n;
  }
  if (S > 1) {     // (also after a failed launch: whatever was queued is joined)
    for (int i = 0; i < S - 1; ++i) {
      HIP_TRY(h, hipEventRecord(h->ev_join[i], h->aux_stream[i]));
      HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_join[i], 0));
    }
  }
  return rc;
}

int32_t pnp_step_row_chunks(const pnp_handle* h, int32_t launches) {
  if (!h || h->newton || !h->have_batch) return 1;
  return step_streams(h, launches);
}

int pnp_integrate(pnp_handle* h, int32_t nt, const int32_t* itout, int32_t n_out, double* cout, int32_t* status) {
  if (!h) return PNP_EINVAL;
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_integrate: call pnp_set_batch first");
  if (nt < 1 || n_out < 0 || (n_out > 0 && (!itout || !cout))) return fail(h, PNP_EINVAL, "pnp_integrate: bad arguments");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const int N = h->a.N, nx = h->a.nx, ldx = h->a.ldx;
  const int64_t B = h->B;
  // CN: for n in range(1,nt) (calculator_old.py:512);  FTCS: for n in range(0,nt) (:990)
  // physical mode: the state after n backward-Euler steps is t_n, like CN
  const int n_first = (h->cfg.method == PNP_METHOD_FTCS) ? 0 : 1;
  int n = n_first;  // index of the next loop pass to run
  for (int io = 0; io <= n_out; ++io) {
    int target;  // run passes n .. target (inclusive)
    if (io < n_out) {
      target = itout[io];
      if (target < n || target > nt - 1) {
        if (target < n_first) continue;  // an output index the loop never visits (e.g. CN n=0)
        return fail(h, PNP_EINVAL, "pnp_integrate: itout must be ascending and < nt");
      }
    } else {
      target = nt - 1;
    }
    const int todo = target - n + 1;
    if (todo > 0) {
      const int rc = pnp_step(h, todo, 0);
      if (rc != PNP_OK) return rc;
      n = target + 1;
    }
    if (io < n_out) {
      double* dst = cout + (size_t)io * B * N * nx;
      HIP_TRY(h, hipMemcpy2DAsync(dst, (size_t)nx * sizeof(double), h->c, (size_t)ldx * sizeof(double), (size_t)nx * sizeof(double),
                                  (size_t)B * N, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (status) return pnp_get_status(h, status);
  return PNP_OK;
}

static int ensure_potential_buffers(pnp_handle* h) {
  if (!h->v) {
    HIP_TRY(h, dev_alloc(h, &h->v, (size_t)h->cfg.batch_capacity * h->a.ldx));
    HIP_TRY(h, dev_alloc(h, &h->gradv, (size_t)h->cfg.batch_capacity * h->a.ldx));
  }
  return PNP_OK;
}

// ode_func (calculator_old.py:827-935) of every lane: f[B][N][ldx] from y[B][N][ldx], both on the device
static int eval_mol_rhs(pnp_handle* h, const double* y, double* f) {
  DevArgs a = h->a;
  if (a.has_rates) {
    a.c = const_cast<double*>(y);   // get_rates(C) of the state being differentiated (calculator_old.py:872-873)
    HIP_TRY(h, launch_rates(a, h->rt, h->rates, h->stream));
  }
  if (waves_per_system(a.nx) > 1) {
    // grids beyond one wave: charge row of the state -> Poisson (multi-wave scans) -> point-wise right-hand side
    const int rc = ensure_potential_buffers(h);
    if (rc != PNP_OK) return rc;
    if (!h->mol_lapl) HIP_TRY(h, dev_alloc(h, &h->mol_lapl, (size_t)h->cfg.batch_capacity * a.ldx));
    DevArgs ay = a;
    ay.c = const_cast<double*>(y);
    if (a.use_mig) {
      HIP_TRY(h, launch_charge_row(ay, h->mol_lapl, h->stream));
      HIP_TRY(h, launch_poisson(a, h->mol_lapl, h->v, h->gradv, h->stream));
    }
    HIP_TRY(h, launch_mol_rhs_pointwise(a, y, h->gradv, f, h->stream));
  } else {
    HIP_TRY(h, launch_mol_rhs(a, y, f, h->stream));
  }
  return PNP_OK;
}

int pnp_mol_rhs(pnp_handle* h, const double* c, double* dcdt) {
  if (!h || !c || !dcdt) return fail(h, PNP_EINVAL, "pnp_mol_rhs: null argument");
  if (h->newton) return fail(h, PNP_EINVAL, "pnp_mol_rhs: not part of the physical mode");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_mol_rhs: call pnp_set_batch first");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const int N = h->a.N, nx = h->a.nx, ldx = h->a.ldx;
  const int64_t B = h->B;
  const size_t cnt = (size_t)h->cfg.batch_capacity * N * ldx;
  if (!h->ytmp) {
    HIP_TRY(h, dev_alloc(h, &h->ytmp, cnt));
    HIP_TRY(h, dev_alloc(h, &h->ftmp, cnt));
    HIP_TRY(h, hipMemsetAsync(h->ytmp, 0, cnt * sizeof(double), h->stream));
  }
  const size_t w = (size_t)nx * sizeof(double), dp = (size_t)ldx * sizeof(double);
  HIP_TRY(h, hipMemcpy2DAsync(h->ytmp, dp, c, w, w, (size_t)B * N, hipMemcpyHostToDevice, h->stream));
  const int rc = eval_mol_rhs(h, h->ytmp, h->ftmp);
  if (rc != PNP_OK) return rc;
  HIP_TRY(h, hipMemcpy2DAsync(dcdt, w, h->ftmp, dp, w, (size_t)B * N, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return PNP_OK;
}

// DOPRI5 (order 5) and DOP853 (order 8) share the call structure: per interval a fresh call (begin, k1, HINIT in the first one),
// then attempted steps in groups of `every` until the counter of lanes still inside the interval reads zero
static int integrate_explicit_rk(pnp_handle* h, const pnp_ode_params* p, int32_t nt, const int32_t* itout, int32_t n_out, double* cout,
                                 int32_t* idid, int64_t* stats, double* t_end, int order) {
  const bool o8 = order == 8;
  const int nbuf = o8 ? 12 : 8;
  if (!h || !p) return fail(h, PNP_EINVAL, "pnp_integrate_dopri5 / _dop853: null argument");
  if (h->newton) return fail(h, PNP_EINVAL, "pnp_integrate_dopri5 / _dop853: not part of the physical mode");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_integrate_dopri5 / _dop853: call pnp_set_batch first");
  if (p->struct_size != (int32_t)sizeof(pnp_ode_params)) return fail(h, PNP_EINVAL, "pnp_integrate_dopri5 / _dop853: struct_size mismatch (ABI)");
  if (nt < 0 || n_out < 0 || (n_out > 0 && (!itout || !cout))) return fail(h, PNP_EINVAL, "pnp_integrate_dopri5 / _dop853: bad output request");
  for (int j = 0; j < n_out; ++j)
    if (itout[j] < 0 || itout[j] >= nt || (j > 0 && itout[j] <= itout[j - 1]))
      return fail(h, PNP_EINVAL, "pnp_integrate_dopri5 / _dop853: itout must be ascending and inside [0, nt)");
  if (p->rtol < 0.0 || p->atol < 0.0 || p->first_step < 0.0 || p->max_step < 0.0 || p->nsteps < 0 || p->nstiff < 0)
    return fail(h, PNP_EINVAL, "pnp_integrate_dopri5 / _dop853: negative parameter");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const int N = h->a.N, nx = h->a.nx, ldx = h->a.ldx;
  const int64_t B = h->B, cap = h->cfg.batch_capacity;
  const size_t cnt = (size_t)cap * N * ldx;
  hipStream_t st = h->stream;
  if (h->ode_buf && h->ode_nbuf < nbuf) {      // a DOPRI5 workspace cannot hold DOP853's twelve buffers
    HIP_TRY(h, hipStreamSynchronize(st));
    (void)hipFree(h->ode_buf);
    h->ode_buf = nullptr;
  }
  if (!h->ode_buf) {
    HIP_TRY(h, dev_alloc(h, &h->ode_buf, (size_t)nbuf * cnt + (size_t)cap * ODE_ND));
    h->ode_nbuf = nbuf;
    HIP_TRY(h, hipMemsetAsync(h->ode_buf, 0, ((size_t)nbuf * cnt + (size_t)cap * ODE_ND) * sizeof(double), st));   // pads of the rows stay zero
  }
  if (!h->ode_int) HIP_TRY(h, dev_alloc(h, &h->ode_int, (size_t)cap * ODE_NI + 64));
  OdeArgs a;
  memset(&a, 0, sizeof(a));
  a.N = N; a.nx = nx; a.ldx = ldx; a.B = B;
  a.nmax = p->nsteps > 0 ? p->nsteps : 500;
  a.nstiff = p->nstiff > 0 ? p->nstiff : 1000;
  a.rtol = p->rtol > 0.0 ? p->rtol : 1e-6;
  a.atol = p->atol > 0.0 ? p->atol : 1e-12;
  a.safe = p->safety > 0.0 ? p->safety : 0.9;
  // scipy's defaults: dopri5 dfactor 0.2 / ifactor 10 / beta 0 -> DOPRI5's 0.04 (< 0: none); dop853 dfactor 0.3 / ifactor 6 / beta 0
  a.facc1 = 1.0 / (p->dfactor > 0.0 ? p->dfactor : (o8 ? 0.3 : 0.2));
  a.facc2 = 1.0 / (p->ifactor > 0.0 ? p->ifactor : (o8 ? 6.0 : 10.0));
  a.beta = o8 ? (p->beta <= 0.0 ? 0.0 : p->beta) : (p->beta == 0.0 ? 0.04 : (p->beta < 0.0 ? 0.0 : p->beta));
  a.expo1 = o8 ? 1.0 / 8.0 - a.beta * 0.2 : 0.2 - a.beta * 0.75;
  a.hinit_expo = o8 ? 1.0 / 8.0 : 1.0 / 5.0;
  a.max_step = p->max_step;
  a.dt = h->a.dt;
  a.y = h->c;
  const int nk = o8 ? 10 : 6;
  for (int j = 0; j < nk; ++j) a.k[j] = h->ode_buf + (size_t)j * cnt;
  a.y1 = h->ode_buf + (size_t)nk * cnt;
  a.ysti = h->ode_buf + (size_t)(nk + 1) * cnt;
  a.d = h->ode_buf + (size_t)nbuf * cnt;
  a.i = h->ode_int;
  a.counters = h->ode_int + (size_t)cap * ODE_NI;
  {   // t = 0, h = first_step, IDID = 1, counters zero
    std::vector<double> d0((size_t)B * ODE_ND, 0.0);
    std::vector<int32_t> i0((size_t)B * ODE_NI, 0);
    for (int64_t b = 0; b < B; ++b) {
      d0[b * ODE_ND + ODE_H] = p->first_step;
      i0[b * ODE_NI + ODE_IDID] = 1;
      i0[b * ODE_NI + ODE_INTERVAL] = -1;
    }
    HIP_TRY(h, hipMemcpyAsync(a.d, d0.data(), d0.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(a.i, i0.data(), i0.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipStreamSynchronize(st));      // the staging vectors go out of scope
  }
  const int every = p->check_every > 0 ? (p->check_every < 32 ? p->check_every : 32) : 4;
  const size_t w = (size_t)nx * sizeof(double), dp = (size_t)ldx * sizeof(double);
  int next_out = 0;
  int64_t step = 0;
  for (int n = 0; n < nt; ++n) {
    a.interval = n;
    HIP_TRY(h, launch_ode_begin(a, st));
    int rc = eval_mol_rhs(h, a.y, a.k[0]);
    if (rc != PNP_OK) return rc;
    if (n == 0 && p->first_step == 0.0) {
      HIP_TRY(h, launch_ode_hinit(a, 0, st));
      rc = eval_mol_rhs(h, a.y1, a.k[1]);
      if (rc != PNP_OK) return rc;
      HIP_TRY(h, launch_ode_hinit(a, 1, st));
    }
    a.slot = (int32_t)(step++ & 63);
    HIP_TRY(h, hipMemsetAsync(a.counters + a.slot, 0, sizeof(int32_t), st));
    HIP_TRY(h, launch_ode_open(a, st));
    // lanes that enter the step loop of this interval: none (every lane has failed in an earlier interval) -> no step group is enqueued
    int32_t left = 1;
    HIP_TRY(h, hipMemcpyAsync(&left, a.counters + a.slot, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    // a lane needs at most nmax + 1 attempted steps per interval; the counter is read every `every` steps
    for (int64_t tries = 0; left != 0 && tries <= (int64_t)a.nmax + 1; tries += every) {
      for (int e = 0; e < every; ++e) {
        a.slot = (int32_t)(step++ & 63);
        if (o8) {
          for (int sgi = 2; sgi <= 12; ++sgi) {
            HIP_TRY(h, launch_ode853_stage(a, sgi, st));
            rc = eval_mol_rhs(h, a.y1, a.k[ode853_stage_dst(sgi)]);
            if (rc != PNP_OK) return rc;
          }
          HIP_TRY(h, launch_ode853_control(a, 0, st));
          rc = eval_mol_rhs(h, a.ysti, a.k[3]);               // f(new state): CALL FCN(N,XPH,K5,K4) of an accepted step
          if (rc != PNP_OK) return rc;
          HIP_TRY(h, hipMemsetAsync(a.counters + a.slot, 0, sizeof(int32_t), st));
          HIP_TRY(h, launch_ode853_control(a, 1, st));
        } else {
          static const int dst[6] = {1, 2, 3, 4, 5, 1};      // k2..k6, then k7 into k2's buffer
          for (int sgi = 2; sgi <= 7; ++sgi) {
            HIP_TRY(h, launch_ode_stage(a, sgi, st));
            rc = eval_mol_rhs(h, sgi == 6 ? a.ysti : a.y1, a.k[dst[sgi - 2]]);
            if (rc != PNP_OK) return rc;
          }
          HIP_TRY(h, hipMemsetAsync(a.counters + a.slot, 0, sizeof(int32_t), st));
          HIP_TRY(h, launch_ode_control(a, st));
        }
      }
      HIP_TRY(h, hipMemcpyAsync(&left, a.counters + a.slot, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(h, hipStreamSynchronize(st));
    }
    if (next_out < n_out && itout[next_out] == n) {
      HIP_TRY(h, hipMemcpy2DAsync(cout + (size_t)next_out * B * N * nx, w, a.y, dp, w, (size_t)B * N, hipMemcpyDeviceToHost, st));
      ++next_out;
    }
  }
  // the last right-hand side of the reference's run is k7 = f(final state): tp.potential / efield belong to the state returned
  // (calculator_old.py:816-818 inside ode_func).  Same here: charge row of the integrated state for pnp_get_state / pnp_get_surface.
  h->cur = 0;
  HIP_TRY(h, launch_charge_row(h->a, h->lapl[0], st));
  h->steps_done = 0;
  std::vector<double> dh((size_t)B * ODE_ND);
  std::vector<int32_t> ih((size_t)B * ODE_NI);
  HIP_TRY(h, hipMemcpyAsync(dh.data(), a.d, dh.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemcpyAsync(ih.data(), a.i, ih.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipStreamSynchronize(st));
  for (int64_t b = 0; b < B; ++b) {
    const int32_t* s = ih.data() + b * ODE_NI;
    if (idid) idid[b] = s[ODE_IDID];
    if (t_end) t_end[b] = dh[b * ODE_ND + ODE_X];
    if (stats) {
      stats[b * 5 + 0] = s[ODE_TOT_NSTEP];
      stats[b * 5 + 1] = s[ODE_TOT_NACCPT];
      stats[b * 5 + 2] = s[ODE_TOT_NREJCT];
      stats[b * 5 + 3] = s[ODE_TOT_NFCN];
      stats[b * 5 + 4] = s[ODE_INTERVAL];
    }
  }
  return PNP_OK;
}

int pnp_integrate_dopri5(pnp_handle* h, const pnp_ode_params* p, int32_t nt, const int32_t* itout, int32_t n_out, double* cout,
                         int32_t* idid, int64_t* stats, double* t_end) {
  return integrate_explicit_rk(h, p, nt, itout, n_out, cout, idid, stats, t_end, 5);
}

int pnp_integrate_dop853(pnp_handle* h, const pnp_ode_params* p, int32_t nt, const int32_t* itout, int32_t n_out, double* cout,
                         int32_t* idid, int64_t* stats, double* t_end) {
  return integrate_explicit_rk(h, p, nt, itout, n_out, cout, idid, stats, t_end, 8);
}

int pnp_integrate_rkc(pnp_handle* h, const pnp_ode_params* p, int32_t nt, const int32_t* itout, int32_t n_out, double* cout,
                      int32_t* idid, int64_t* stats, double* t_end) {
  if (!h || !p) return fail(h, PNP_EINVAL, "pnp_integrate_rkc: null argument");
  if (h->newton) return fail(h, PNP_EINVAL, "pnp_integrate_rkc: not part of the physical mode");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_integrate_rkc: call pnp_set_batch first");
  if (p->struct_size != (int32_t)sizeof(pnp_ode_params)) return fail(h, PNP_EINVAL, "pnp_integrate_rkc: struct_size mismatch (ABI)");
  if (nt < 0 || n_out < 0 || (n_out > 0 && (!itout || !cout))) return fail(h, PNP_EINVAL, "pnp_integrate_rkc: bad output request");
  for (int j = 0; j < n_out; ++j)
    if (itout[j] < 0 || itout[j] >= nt || (j > 0 && itout[j] <= itout[j - 1]))
      return fail(h, PNP_EINVAL, "pnp_integrate_rkc: itout must be ascending and inside [0, nt)");
  if (p->rtol < 0.0 || p->atol < 0.0 || p->max_step < 0.0 || p->nsteps < 0) return fail(h, PNP_EINVAL, "pnp_integrate_rkc: negative parameter");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const int N = h->a.N, nx = h->a.nx, ldx = h->a.ldx;
  const int64_t B = h->B, cap = h->cfg.batch_capacity;
  const size_t cnt = (size_t)cap * N * ldx;
  hipStream_t st = h->stream;
  const int nbuf = 8;      // (DOPRI5's workspace: five of its buffers are used here)
  if (!h->ode_buf) {
    HIP_TRY(h, dev_alloc(h, &h->ode_buf, (size_t)nbuf * cnt + (size_t)cap * ODE_ND));
    h->ode_nbuf = nbuf;
    // the whole workspace once (a later pnp_integrate_dopri5 on this handle finds it allocated and relies on zero row pads everywhere)
    HIP_TRY(h, hipMemsetAsync(h->ode_buf, 0, ((size_t)nbuf * cnt + (size_t)cap * ODE_ND) * sizeof(double), st));
  }
  HIP_TRY(h, hipMemsetAsync(h->ode_buf, 0, (size_t)5 * cnt * sizeof(double), st));   // pads of the rows stay zero
  if (!h->rkc_d) HIP_TRY(h, dev_alloc(h, &h->rkc_d, (size_t)cap * RKC_ND));
  if (!h->rkc_i) HIP_TRY(h, dev_alloc(h, &h->rkc_i, (size_t)cap * RKC_NI + 64));
  RkcArgs a;
  memset(&a, 0, sizeof(a));
  a.N = N; a.nx = nx; a.ldx = ldx; a.B = B;
  a.nmax = p->nsteps > 0 ? p->nsteps : 100000;
  a.rtol = p->rtol > 0.0 ? p->rtol : 1e-6;
  a.atol = p->atol > 0.0 ? p->atol : 1e-12;
  {
    const double mm = std::sqrt(a.rtol / (10.0 * 2.22e-16));      // beyond this many stages the recurrence amplifies round-off (rkc.f)
    a.mmax = (int32_t)std::llround(mm < 2.0 ? 2.0 : (mm > 1.0e6 ? 1.0e6 : mm));
  }
  a.max_step = p->max_step;
  a.dt = h->a.dt;
  a.y = h->c;
  a.fn = h->ode_buf;
  a.F = h->ode_buf + cnt;
  a.arg = h->ode_buf + 2 * cnt;
  a.yjm2 = h->ode_buf + 3 * cnt;
  a.ev = h->ode_buf + 4 * cnt;
  a.d = h->rkc_d;
  a.i = h->rkc_i;
  a.counters = h->rkc_i + (size_t)cap * RKC_NI;
  HIP_TRY(h, hipMemsetAsync(a.d, 0, (size_t)cap * RKC_ND * sizeof(double), st));
  HIP_TRY(h, hipMemsetAsync(a.i, 0, ((size_t)cap * RKC_NI + 64) * sizeof(int32_t), st));
  {   // IDID = 1
    std::vector<int32_t> i0((size_t)B * RKC_NI, 0);
    for (int64_t b = 0; b < B; ++b) {
      i0[b * RKC_NI + RKI_IDID] = 1;
      i0[b * RKC_NI + RKI_INTERVAL] = -1;
    }
    HIP_TRY(h, hipMemcpyAsync(a.i, i0.data(), i0.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipStreamSynchronize(st));
  }
  const int every = p->check_every > 0 ? (p->check_every < 32 ? p->check_every : 32) : 16;
  const size_t w = (size_t)nx * sizeof(double), dp = (size_t)ldx * sizeof(double);
  int next_out = 0;
  int64_t tick = 0;
  bool budget_exhausted = false;
  for (int n = 0; n < nt; ++n) {
    a.interval = n;
    HIP_TRY(h, launch_rkc_begin(a, st));
    a.slot = (int32_t)(tick++ & 63);
    HIP_TRY(h, hipMemsetAsync(a.counters + a.slot, 0, sizeof(int32_t), st));   // (the ticks below clear each other's slots)
    HIP_TRY(h, launch_rkc_advance(a, st));      // first call: argument = the state; later calls: the next step's first stage
    int32_t left = 1;
    HIP_TRY(h, hipMemcpyAsync(&left, a.counters + a.slot, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));       // (no lane left -- all failed earlier: no tick is enqueued)
    // a lane needs at most nmax attempted steps of at most mmax + 1 ticks, plus the power iterations (<= 50 per estimate)
    const int64_t limit = ((int64_t)a.nmax + 1) * ((int64_t)a.mmax + 53) + 64;
    for (int64_t tries = 0; left != 0 && tries <= limit; tries += every) {
      for (int e = 0; e < every; ++e) {
        const int rc = eval_mol_rhs(h, a.arg, a.F);
        if (rc != PNP_OK) return rc;
        a.slot = (int32_t)(tick++ & 63);       // (cleared by the advance kernel of the tick before)
        HIP_TRY(h, launch_rkc_advance(a, st));
      }
      HIP_TRY(h, hipMemcpyAsync(&left, a.counters + a.slot, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(h, hipStreamSynchronize(st));
    }
    if (left != 0) {        // the tick budget ran out with lanes still inside the interval: they are reported as -2 below, nothing is output
      budget_exhausted = true;
      break;
    }
    if (next_out < n_out && itout[next_out] == n) {
      HIP_TRY(h, hipMemcpy2DAsync(cout + (size_t)next_out * B * N * nx, w, a.y, dp, w, (size_t)B * N, hipMemcpyDeviceToHost, st));
      ++next_out;
    }
  }
  h->cur = 0;
  HIP_TRY(h, launch_charge_row(h->a, h->lapl[0], st));
  h->steps_done = 0;
  std::vector<double> dh((size_t)B * RKC_ND);
  std::vector<int32_t> ih((size_t)B * RKC_NI);
  HIP_TRY(h, hipMemcpyAsync(dh.data(), a.d, dh.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipMemcpyAsync(ih.data(), a.i, ih.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipStreamSynchronize(st));
  for (int64_t b = 0; b < B; ++b) {
    const int32_t* s = ih.data() + b * RKC_NI;
    if (idid) idid[b] = (budget_exhausted && s[RKI_ACTIVE] != 0) ? -2 : s[RKI_IDID];
    if (t_end) t_end[b] = dh[b * RKC_ND + RKC_T];
    if (stats) {
      stats[b * 7 + 0] = s[RKI_TOT_NSTEP];
      stats[b * 7 + 1] = s[RKI_TOT_NACCPT];
      stats[b * 7 + 2] = s[RKI_TOT_NREJCT];
      stats[b * 7 + 3] = s[RKI_TOT_NFE];
      stats[b * 7 + 4] = s[RKI_INTERVAL];
      stats[b * 7 + 5] = s[RKI_TOT_NFESIG];
      stats[b * 7 + 6] = s[RKI_MAXM];
    }
  }
  return PNP_OK;
}

int pnp_get_state(pnp_handle* h, double* c, double* v, double* grad_v, double* lapl_v) {
  if (!h) return PNP_EINVAL;
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_get_state: call pnp_set_batch first");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const int N = h->a.N, nx = h->a.nx, ldx = h->a.ldx;
  const int64_t B = h->B;
  const size_t w = (size_t)nx * sizeof(double), dp = (size_t)ldx * sizeof(double);
  if (c) HIP_TRY(h, hipMemcpy2DAsync(c, w, h->c, dp, w, (size_t)B * N, hipMemcpyDeviceToHost, h->stream));
  if (h->newton) {
    // the potential is part of the state; its gradient (centred, one-sided at the ends) and -rho/eps are derived here
    std::vector<double> ph, cc;
    if (v || grad_v) {
      ph.resize((size_t)B * nx);
      HIP_TRY(h, hipMemcpy2DAsync(ph.data(), w, h->v, dp, w, (size_t)B, hipMemcpyDeviceToHost, h->stream));
    }
    if (lapl_v) {
      cc.resize((size_t)B * N * nx);
      HIP_TRY(h, hipMemcpy2DAsync(cc.data(), w, h->c, dp, w, (size_t)B * N, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int64_t b = 0; b < B; ++b) {
      if (v) memcpy(v + (size_t)b * nx, ph.data() + (size_t)b * nx, w);
      if (grad_v) {
        const double* p = ph.data() + (size_t)b * nx;
        double* g = grad_v + (size_t)b * nx;
        const std::vector<double>& xg = h->xgrid;
        for (int i = 1; i < nx - 1; ++i) g[i] = (p[i + 1] - p[i - 1]) / (xg[i + 1] - xg[i - 1]);
        g[0] = (p[1] - p[0]) / (xg[1] - xg[0]);
        g[nx - 1] = (p[nx - 1] - p[nx - 2]) / (xg[nx - 1] - xg[nx - 2]);
      }
      if (lapl_v)
        for (int i = 0; i < nx; ++i) {
          double rho = 0;
          for (int k = 0; k < N; ++k) rho += h->qk[k] * cc[((size_t)b * N + k) * nx + i];
          lapl_v[(size_t)b * nx + i] = -rho / h->a.eps;
        }
    }
    return PNP_OK;
  }
  // Poisson solve of the most recent step used the lagged row; before any step it is the initial row
  const double* lagged = (h->steps_done > 0) ? h->lapl[1 - h->cur] : h->lapl[h->cur];
  if (v || grad_v) {
    const int rc = ensure_potential_buffers(h);
    if (rc != PNP_OK) return rc;
    HIP_TRY(h, launch_poisson(h->a, lagged, h->v, h->gradv, h->stream));
    if (v) HIP_TRY(h, hipMemcpy2DAsync(v, w, h->v, dp, w, (size_t)B, hipMemcpyDeviceToHost, h->stream));
    if (grad_v) HIP_TRY(h, hipMemcpy2DAsync(grad_v, w, h->gradv, dp, w, (size_t)B, hipMemcpyDeviceToHost, h->stream));
  }
  if (lapl_v) HIP_TRY(h, hipMemcpy2DAsync(lapl_v, w, lagged, dp, w, (size_t)B, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return PNP_OK;
}

int pnp_get_surface(pnp_handle* h, double* csurf, double* vsurf, double* esurf) {
  if (!h) return PNP_EINVAL;
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_get_surface: call pnp_set_batch first");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const int64_t B = h->B;
  const int ldx = h->a.ldx;
  if (csurf) {
    HIP_TRY(h, launch_surface(h->a, h->csurf, h->stream));
    HIP_TRY(h, hipMemcpyAsync(csurf, h->csurf, (size_t)B * h->a.N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  if (h->newton && (vsurf || esurf)) {
    std::vector<double> p01((size_t)B * 2);
    HIP_TRY(h, hipMemcpy2DAsync(p01.data(), 2 * sizeof(double), h->v, (size_t)ldx * sizeof(double), 2 * sizeof(double), (size_t)B,
                                hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int64_t b = 0; b < B; ++b) {
      if (vsurf) vsurf[b] = p01[2 * b];
      if (esurf) esurf[b] = -(p01[2 * b + 1] - p01[2 * b]) / (h->xgrid[1] - h->xgrid[0]);   // the one-sided field of the Stern condition
    }
    return PNP_OK;
  }
  if (vsurf || esurf) {
    const int rc = ensure_potential_buffers(h);
    if (rc != PNP_OK) return rc;
    const double* lagged = (h->steps_done > 0) ? h->lapl[1 - h->cur] : h->lapl[h->cur];
    HIP_TRY(h, launch_poisson(h->a, lagged, h->v, h->gradv, h->stream));
    const size_t dp = (size_t)ldx * sizeof(double);
    if (vsurf) HIP_TRY(h, hipMemcpy2DAsync(vsurf, sizeof(double), h->v, dp, sizeof(double), (size_t)B, hipMemcpyDeviceToHost, h->stream));
    if (esurf) HIP_TRY(h, hipMemcpy2DAsync(esurf, sizeof(double), h->gradv, dp, sizeof(double), (size_t)B, hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (esurf)
    for (int64_t b = 0; b < B; ++b) esurf[b] = -esurf[b];  // tp.efield = -grad_v, calculator_old.py:816
  return PNP_OK;
}

int pnp_get_status(pnp_handle* h, int32_t* status) {
  if (!h || !status) return fail(h, PNP_EINVAL, "pnp_get_status: null argument");
  if (!h->have_batch) return fail(h, PNP_ESTATE, "pnp_get_status: call pnp_set_batch first");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(status, h->status, (size_t)h->B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return PNP_OK;
}

int pnp_synchronize(pnp_handle* h) {
  if (!h) return PNP_EINVAL;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return PNP_OK;
}

int pnp_timer_start(pnp_handle* h) {
  if (!h) return PNP_EINVAL;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
  return PNP_OK;
}

int pnp_timer_stop(pnp_handle* h, float* elapsed_ms) {
  if (!h || !elapsed_ms) return fail(h, PNP_EINVAL, "pnp_timer_stop: null argument");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
  HIP_TRY(h, hipEventSynchronize(h->ev1));
  HIP_TRY(h, hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
  return PNP_OK;
}

int64_t pnp_device_bytes(const pnp_handle* h) { return h ? h->dev_bytes : 0; }
int32_t pnp_row_pitch(const pnp_handle* h) { return h ? h->a.ldx : 0; }

}  // extern "C"
