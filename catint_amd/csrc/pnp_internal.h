// Internal declarations shared by the HIP kernels (pnp_kernels.hip) and the C-ABI
// implementation (pnp_capi.hip).  gfx950 / MI355X only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/catint_pnp.h"

namespace pnp {

// Debug / tuning options of a handle (C-ABI: pnp_set_option).  The CATINT_* environment variables of the same names are read ONCE,
// in pnp_create, as the defaults of a new handle; nothing in the library reads the environment afterwards.
// (NK_WORKGROUP: the library's own choice among the workgroup-per-point / team / sweep kernels, lane kernels excluded)
enum NewtonKernelChoice { NK_AUTO = 0, NK_GENERIC, NK_TEAM, NK_SWEEP, NK_BOTH, NK_LANE, NK_LANE2, NK_LANE4, NK_WORKGROUP };
struct Options {
  int newton_kernel = NK_AUTO;       // NEWTON_KERNEL = generic | team | sweep | both | lane | lane2 | lane4 | workgroup (tests, pnp_autotune)
  int lane_records_f32 = 0;          // LANE_RECORDS = f32: the fused lane kernel keeps the columns T of its records in single precision (N >= 2)
  int lane_fused = -1;               // LANE_FUSED = 0 | 1: the lane kernel with the update inside the back-substitution; -1: by the batch
  int newton_exchange_global = 0;    // NEWTON_EXCHANGE = global: the row-per-thread kernel's buffers in device memory (creation time)
  int newton_team_threads = 0;       // NEWTON_TEAM_THREADS = 256 | 512 | 1024
  int newton_regs = 0;               // NEWTON_REGS = 512: the 512-register build of the row-per-thread kernel (N = 5, 6)
  int newton_blocks = 0;             // NEWTON_BLOCKS: size of the persistent grid of the workgroup-per-point kernels
  int newton_lane_groups = 0;        // NEWTON_LANE_GROUPS: groups the lane kernels' workspace holds (tests: several chunks)
  int newton_sweep_blocks = 0;       // NEWTON_SWEEP_BLOCKS: waves the sweep kernels' workspace holds
  double lane_pivot_limit = 0.0;     // LANE_PIVOT_LIMIT: pivot monitor of the lane kernels (0: PIVOT_GROWTH_LIMIT)
  int lane_order = 1;                // LANE_ORDER = 0: slot s holds operating point s
  int lane_stagger = -1;             // LANE_STAGGER: start delay between the four phase groups of the lane-quad kernel's waves, in units of
                                     // ~4 us (s_sleep 127); -1: the kernel's own choice, 0: none
  int pnp_kernel = 0;                // PNP_KERNEL = 2 (LDS-staged step_kernel) | 4 (register-resident) | 5, 6, 7 (streaming)
  int pnp_waves_per_grid = 0;        // PNP_WAVES_PER_GRID
  int pnp_species_per_wave = 0;      // PNP_SPECIES_PER_WAVE
  int pnp_step_streams = 0;          // PNP_STEP_STREAMS: row chunks of a pnp_step call of several launches
  int pnp_alternate_rows = -1;       // PNP_ALTERNATE_ROWS = 0 | 1; -1: by the size of the state
  int pnp_st_waves_per_cu = 0;       // PNP_ST_WAVES_PER_CU: resident waves of the streaming kernel
  int pnp_no_post_upload_dispatch = 0;   // PNP_NO_POST_UPLOAD_DISPATCH = 1 (probes)
};

// Per-species constants, computed once on the host with the reference's expression order
// (catint/calculator_old.py:543-547, :1013-1017; catint/transport.py:436) and read by the kernels
// through scalar loads.
struct SpecConst {
  double D;      // tp.D[k]
  double q;      // tp.charges[k] = z*F
  double mu;     // D*q*beta                         transport.py:436
  double s;      // D*dt/dx**2 (+0.5 with --LF)      calculator_old.py:543-546
  double hs;     // 0.5*s
  double ee;     // q*beta*dt*D                      :547
  double e4;     // ee/4/dx      (add_field :486-490)
  double rdiag;  // 1/(1+s)
  double oms;    // 1-s
  double qe;     // q/eps        (:770)
  double twoD;   // 2*D
  double sf;     // D*dt/dx**2   (FTCS :1013)
  double dm;     // dt/(2*dx)*mu (FTCS :1014)
  double Mf;     // FTCS centre weight after the LF handling (:1015, :1021)
  // the same CN constants divided by the diagonal 1+s (rows are solved with unit diagonal)
  double hsr, e4r, eer, omsr;
  double pad0, pad1;
};

// Everything a workgroup needs to advance one operating point; passed by value as the kernel argument.
struct DevArgs {
  int32_t N;         // species
  int32_t nx;        // grid points incl. the two boundary points
  int32_t m;         // nx-2 interior unknowns of every tridiagonal system
  int32_t ldx;       // row pitch in doubles (multiple of 16 -> rows are 128-B aligned)
  int32_t pb_mode;
  int32_t method;
  int32_t lf;
  int32_t use_mig;
  int32_t nsteps;    // timesteps fused into this launch
  int32_t has_rates; // FTCS: add rates[b][k][i]*dt (computed by rates_kernel before the step)
  int32_t st_waves_per_cu;   // streaming kernel: resident waves per CU (Options::pnp_st_waves_per_cu; 0: what fits)
  int32_t reverse;   // streaming kernel: walk the operating points from the last to the first (see pnp_step: alternate launches of one
                     // timestep start on the rows the previous launch wrote last, which the caches still hold)
  int64_t B;
  double dx, dt, beta, eps;
  double dx2, inv2dx, nxm1;   // dx*dx, 1/(2 dx), (double)(nx-1): wave-uniform fp64 values would otherwise live in (spilled) vector registers
  const SpecConst* spec;  // [N]
  // state
  double* c;          // [B][N][ldx]  concentrations, in place
  double* lapl_a;     // [B][ldx]     -sum_k q_k c_k / eps: read by the first step of this launch
  double* lapl_b;     // [B][ldx]     written by the first step (ping-pong with lapl_a)
  const double* rates;  // [B][N][ldx] or null
  // per lane
  const double* pb;     // [B][4]
  const double* vzeta;  // [B]
  const double* flux;   // [B][N]
  const double* cbulk;  // [B][N]   C0[k][nx-1]
  int32_t* status;      // [B]
};

struct ReactionTable {
  int32_t n;
  int32_t n_lhs[PNP_MAX_REACTIONS];
  int32_t n_rhs[PNP_MAX_REACTIONS];
  int32_t lhs[PNP_MAX_REACTIONS][PNP_MAX_REACTANTS];
  int32_t rhs[PNP_MAX_REACTIONS][PNP_MAX_REACTANTS];
  double kf[PNP_MAX_REACTIONS];
  double kr[PNP_MAX_REACTIONS];
};

// The same table flattened for the lane kernels (pnp_lane*.hip): one entry per reaction SIDE that carries a rate, with everything the
// assembly needs as plain numbers -- no index lists, no trip counts, nothing to branch on.  Side s contributes
//   rate_s = k_s gam^order_s prod_a c_(i_a)           (a = 0 .. 3: the reactants of that side, repeats allowed; `slots` holds for each a
//                                                      the row of the kernel's per-row value table in bits 4a .. 4a+3 -- the species,
//                                                      or 8 = the constant one for an unused slot -- and the species column the
//                                                      derivative goes to in bits 16+4a .. 16+4a+3, 15 = none)
// to R_k with the weight w_s[k] = +-(occurrences of k among the products - occurrences among the educts), the sign taking care of
// "forward minus backward".  Built on the host by pnp_set_reactions, copied into LDS once per kernel.
struct ReactionSides {
  int32_t n;                  // sides (even: padded with a side without a rate)
  int32_t max_exponent;       // largest multiplicity of one species on one side (the lane kernels evaluate up to 2)
  struct Side {
    double k;
    double w[PNP_NEWTON_MAX_SPECIES];
    int32_t order;            // reactants on this side (power of the activity coefficient)
    uint32_t slots;
  } side[2 * PNP_MAX_REACTIONS];
};

// points-per-lane template instance that covers nx (interior m = nx-2 <= 64*P*waves_per_system); 0 if unsupported
int points_per_lane(int nx);
// waves cooperating on one tridiagonal system: 1 up to nx = 1026, 2 up to 2050, 4 up to 4098
int waves_per_system(int nx);
hipError_t launch_step_mw(const DevArgs& a, hipStream_t stream);
// W = waves per operating point, G = species interleaved inside one wave, chosen for a batch of B lanes and for launches
// of one timestep or of many (fused)
void choose_step_config(int N, int64_t B, int P, bool fused, int* W, int* G);
bool step_config_supported(int W, int G);
size_t step_lds_bytes(int P, int W, int G);

// launchers (all asynchronous on `stream`)
hipError_t launch_step(const DevArgs& a, int W, int G, hipStream_t stream);
// register-resident kernel (step_kernel_rr): W species-parallel waves per operating point;
// applicable to the Dirichlet/Dirichlet Poisson branch with an even number of points per lane
bool step_rr_applicable(const DevArgs& a);
hipError_t launch_step_rr(const DevArgs& a, int W, hipStream_t stream);
// streaming kernel (pnp_stream.hip: step_kernel_st): persistent single-wave workgroups, next row prefetched behind the current
// solve; same applicability as the register-resident kernel.  mode 0: registers only, 1: charge / gradient rows of the step in LDS
hipError_t launch_step_st(const DevArgs& a, int mode, hipStream_t stream);
// lapl[b][i] = -sum_k q_k c[b][k][i]/eps for all nx points (initial charge row, calculator_old.py:767-771)
hipError_t launch_charge_row(const DevArgs& a, double* lapl, hipStream_t stream);
// v, grad_v [B][ldx] from a lapl row (get_potential_and_gradient, calculator_old.py:773-803)
hipError_t launch_poisson(const DevArgs& a, const double* lapl, double* v, double* gradv, hipStream_t stream);
// rates[b][k][i] (get_rates, calculator_old.py:159-208)
hipError_t launch_rates(const DevArgs& a, const ReactionTable& rt, double* rates, hipStream_t stream);
// method-of-lines RHS (ode_func, calculator_old.py:827-935): dydt[b][k][i] from y[b][k][i]
hipError_t launch_mol_rhs(const DevArgs& a, const double* y, double* dydt, hipStream_t stream);
// the same for grids spanning several waves: point-wise from a gradient row computed by launch_poisson
hipError_t launch_mol_rhs_pointwise(const DevArgs& a, const double* y, const double* gradv, double* dydt, hipStream_t stream);
// state upload: contiguous staging buffer [B][N][nx] -> pitched rows a.c (pads zero), cbulk[B][N], zeroed charge row (nullable),
// status[B] = 0, iters[B] = 0 (nullable)
hipError_t launch_unpack_state(const DevArgs& a, const double* stage, double* cbulk, double* lapl_zero, int32_t* iters_zero,
                               hipStream_t stream);
// surface gather: csurf[B][N] = c[b][k][0]
hipError_t launch_surface(const DevArgs& a, double* csurf, hipStream_t stream);

// ---- physical mode (pnp_newton.hip): fully implicit coupled Newton, block-tridiagonal PCR ------------------------
struct NewtonArgs {
  int32_t N, nx, ldx, nsteps;
  int32_t maxit, wall_bc, mpb, RS;       // RS: row stride of the element-major PCR buffers
  int32_t estimate, stationary;          // accept on the quadratic error estimate (pnp_newton_params.error_estimate); 1/dt = 0 (one solve)
  int64_t B;
  int64_t work_stride;                   // doubles per workgroup in `work`
  double tol, dphi_max;
  double stern;                          // dx*C_S/eps              (Stern Robin wall, comsol_model.py:613,:982)
  double phi_pzc;
  double vt_inv;                         // beta*max(|q|max, 1): 1/thermal voltage of the highest valence
  double qb[PNP_NEWTON_MAX_SPECIES];     // q_k*beta
  double sig[PNP_NEWTON_MAX_SPECIES];    // dx^2/(D_k dt), 0 for the stationary problem
  double fl[PNP_NEWTON_MAX_SPECIES];     // dx/D_k      (scales the wall flux)
  double peq[PNP_NEWTON_MAX_SPECIES];    // dx^2/eps*q_k
  double vol[PNP_NEWTON_MAX_SPECIES];    // N_A a_k^3   (MPB, comsol_model.py:1041-1063)
  double rs[PNP_NEWTON_MAX_SPECIES];     // dx^2/D_k    (scales the reaction source)
  double pe[PNP_NEWTON_MAX_SPECIES];     // v dx/D_k    (constant convection velocity v, pnp_set_convection; 0 without)
  int32_t convect, pad4_;                // v != 0
  const struct ReactionTable* rt;        // device copy of the mass-action table, or null
  const struct ReactionSides* sides;     // device copy of the flattened table (lane kernels), or null
  int32_t n_wk;                          // first-order surface reactions (pnp_set_wall_kinetics)
  int32_t wk_species[PNP_MAX_WALL_REACTIONS];                     // species whose surface concentration enters, -1: zeroth order
  double wk_nu[PNP_MAX_WALL_REACTIONS][PNP_NEWTON_MAX_SPECIES];   // stoichiometry of the flux INTO the domain
  const double* wk_k;                    // [B][PNP_MAX_WALL_REACTIONS] rate constants per lane
  double wk_alpha[PNP_MAX_WALL_REACTIONS];   // Butler-Volmer exponent [1/V]: rate *= exp(alpha (phiM - phi(x=0)))  (pnp_set_wall_rate_law)
  double wk_sat[PNP_MAX_WALL_REACTIONS];     // Langmuir saturation [m^3/mol]: c_s -> c_s/(1 + K_sat c_s)
  const double* gw;                      // [nx] grid: dx/h_e of edge e (points e, e+1); 1 on the uniform grid
  const double* gv;                      // [nx] grid: control volume V_i/dx; 1 inside a uniform grid, 1/2 at the ends
  double* c;                             // [B][N][ldx] state = Newton iterate, in place
  double* c_old;                         // [B][N][ldx] previous time level
  double* phi;                           // [B][ldx]
  const double* pb;                      // [B][4]: wall potential phiM, bulk potential
  const double* flux;                    // [B][N] wall flux INTO the domain
  const double* cbulk;                   // [B][N]
  double* work;                          // PCR exchange buffers in device memory, or null -> dynamic LDS
  double* stash;                         // pair kernel: parked a-rows, stash_stride doubles per workgroup
  int64_t stash_stride;
  int32_t* status;                       // [B]
  int32_t* iters;                        // [B] Newton iterations spent by this call (maxit+1 for a failed solve)
  const int32_t* lane_mask;              // [B] or null: lanes with a zero are left untouched (frozen lanes of the SCF loop)
  double* sweep;                         // sweep kernel (one team per operating point): records, sweep_stride doubles per team
  int64_t sweep_stride;
  int32_t sweep_blocks, pad2_;           // workgroups (= waves) the sweep workspace was sized for; 0: no workspace
  // lane kernel (pnp_lane.hip: one operating point per lane): batch-innermost copies of the state and the block-Thomas records
  double* lane_ts;                       // [groups][nx][(N+2)/2][32][2] concentrations + potential of 32 operating points
  double* lane_xs;                       // [groups][nx][(N+2)/2][32][2] Newton update
  double* lane_tco;                      // [groups][nx][(N+1)/2][32][2] previous time level
  double* lane_rec;                      // [groups][nx][((N+1)^2 + (N+1))/2][32][2]
  double* lane_tcn;                      // [groups][nx][(N+1)/2][32][2] BDF2 inside a launch: the time level before the previous one
  double* c_old2;                        // [B][N][ldx] its home between launches (null without BDF2)
  int32_t bdf2, bdf_hist0;               // lane kernels: BDF2 steps inside the launch; c_old2 holds a history at its start
  int64_t lane_groups;                   // groups (of 32 operating points) the three buffers hold
  int64_t lane_group0;                   // first group of this launch (the batch is walked in chunks of lane_groups)
  const int32_t* lane_perm;              // [B] operating point of slot s (slot = group * points per group + lane), or null: slot s holds
                                         // point s.  The host orders the points by expected Newton iterations (pnp_capi.hip:
                                         // lane_order) so that the lanes of a wave finish together.
  int32_t lane_lg, pad3_;                // operating points per group: 32 (lane kernel) or 16 (lane-pair kernel, pnp_lane2.hip)
  double lane_pivot_limit;               // pivot monitor of the lane kernels (pnp_lane_common.h): multipliers beyond this mark the lane
  const Options* opt;                    // HOST pointer (the launchers' kernel choice); never dereferenced on the device
  int32_t lane_stagger;
  int32_t ext_old;                       // 1: c_old holds the previous-level combination of this (single) step, prepared by the caller
                                         // (BDF2: (4 c_n - c_n-1)/3, with sig scaled by 3/2); the kernels do not overwrite it           // lane-quad kernel: wave g starts (g & 3) * lane_stagger sleep periods late (see pnp_lane4.hip)
};
int newton_threads(int nb, int nx);
size_t newton_exchange_doubles(int nb, int nx);
size_t newton_team_doubles(int nb, int nx);      // row buffer of the lane-team kernel (N >= 5)
size_t newton_sweep_doubles(int nb, int nx);     // records of one team of the sweep kernel (block Thomas, large batches)
bool newton_sweep_preferred(int nb, int nx, int64_t B, int mode, const Options& opt);
bool newton_sweep_two_sided(int nb, int nx, int64_t B, int mode, const Options& opt);    // ... with two teams per operating point (elimination from both ends)   // large blocks and enough lanes to fill the chip with teams (mode: 0 point ions, 1 steric, 2 + reactions)
bool newton_exchange_in_lds(int nb, int nx, const Options& opt);
int newton_pair_threads(int nb, int nx);   // threads of the pair kernel, 0 if the shape does not fit it
int newton_pair_stride(int nb, int nx);    // its compile-time row stride (256 or 512)
hipError_t launch_newton(const NewtonArgs& a, int blocks, hipStream_t stream);
// lane kernel (pnp_lane.hip): one operating point per lane, block Thomas from both ends in registers; point / steric ions without
// homogeneous reactions
bool newton_lane_supported(int nb, int nx, int mode);
bool newton_lane_preferred(int nb, int nx, int64_t B, int mode, const Options& opt);
size_t newton_lane_rec_doubles(int nb, int nx);       // records of one group of 32 operating points
size_t newton_lane_state_doubles(int nb, int nx);     // transposed state + previous time level of one group
hipError_t launch_newton_lane(const NewtonArgs& a, hipStream_t stream);
hipError_t launch_lane_transpose(const NewtonArgs& a, int64_t ngroups, bool in, hipStream_t stream);     // a.lane_lg points per group
// lane-pair kernel (pnp_lane2.hip): four lanes per operating point (two directions x two halves of the block row), N >= 5
bool newton_lane2_supported(int nb, int nx, int mode);
bool newton_lane2_preferred(int nb, int nx, int64_t B, int mode, const Options& opt);
size_t newton_lane2_rec_doubles(int nb, int nx);      // per group of 16 operating points
size_t newton_lane2_state_doubles(int nb, int nx);
hipError_t launch_newton_lane2(const NewtonArgs& a, hipStream_t stream);
// lane-quad kernel (pnp_lane4.hip): eight lanes per operating point (two directions x four column lanes of the block row), N >= 5
bool newton_lane4_supported(int nb, int nx, int mode);
bool newton_lane4_preferred(int nb, int nx, int64_t B, int mode, const Options& opt);
size_t newton_lane4_rec_doubles(int nb, int nx);      // per group of 8 operating points
size_t newton_lane4_state_doubles(int nb, int nx);
hipError_t launch_newton_lane4(const NewtonArgs& a, hipStream_t stream);

// ---- kinetics <-> transport SCF loop on the device (pnp_scf.hip; Calculator.run_scf_cycle, calculator.py:294-406) ----
struct ScfArgs {
  int32_t N, n_wk, iH, iOH;
  int32_t istep, ldx, slot, pad_;
  int64_t B;
  double tau, h0, faraday;               // convergence threshold tau_scf; first cell x[1]-x[0]; unit_F of the caller
  int32_t wk_species[PNP_MAX_WALL_REACTIONS];
  double wk_nu[PNP_MAX_WALL_REACTIONS][PNP_NEWTON_MAX_SPECIES];
  double nel[PNP_NEWTON_MAX_SPECIES], nprod[PNP_NEWTON_MAX_SPECIES];
  const double* wk_k;                    // [B][PNP_MAX_WALL_REACTIONS]
  double wk_alpha[PNP_MAX_WALL_REACTIONS], wk_sat[PNP_MAX_WALL_REACTIONS];     // rate law (pnp_set_wall_rate_law), see NewtonArgs
  const double* pb;                      // [B][4]: wall potential phiM first
  double *sc, *sc_old, *flux, *cd_old;   // [B][N]
  double *mix, *acc, *surface_pH, *vsurf, *esurf;      // [B]
  int32_t *step_to_check, *active, *failed;            // [B]
  double* c;                             // [B][N][ldx] transport state
  double* phi;                           // [B][ldx]
  double* snap_c;                        // the state each lane's last CONVERGED transport solve left behind ...
  double* snap_phi;                      // ... restored when a solve fails
  const int32_t* status;                 // [B] of the transport solve
  int32_t* counters;                     // [0..63] active lanes after iteration (slot = istep & 63), [64] last iteration with work
};
hipError_t launch_scf_pre(const ScfArgs& a, hipStream_t stream);     // mixing, fallback, pH, new wall fluxes
hipError_t launch_scf_keep(const ScfArgs& a, hipStream_t stream);    // converged lanes: state -> snapshot; failed lanes: snapshot -> state
hipError_t launch_scf_post(const ScfArgs& a, hipStream_t stream);    // surface state, accuracy, convergence flags

// ---- method-of-lines integrator on the device (pnp_ode.hip; scipy 'dopri5' = Hairer's DOPRI5, calculator_old.py:955-963) ----
enum { ODE_X = 0, ODE_H, ODE_XEND, ODE_FACOLD, ODE_HLAMB, ODE_HMAX, ODE_ERR, ODE_HTRY, ODE_DNF, ODE_HNEW, ODE_ND = 10 };
enum { ODE_ACTIVE = 0, ODE_LAST, ODE_REJECT, ODE_NSTEP, ODE_NACCPT, ODE_IASTI, ODE_NONSTI, ODE_IDID, ODE_INTERVAL,
       ODE_TOT_NSTEP, ODE_TOT_NACCPT, ODE_TOT_NREJCT, ODE_TOT_NFCN, ODE_PENDING, ODE_NI = 14 };
struct OdeArgs {
  int32_t N, nx, ldx, nmax;
  int32_t nstiff, slot, interval, pad_;
  int64_t B;
  double rtol, atol, safe, facc1, facc2, beta, expo1, max_step, dt;
  double hinit_expo;  // HINIT: 1/IORD (1/5 DOPRI5, 1/8 DOP853)
  double* y;          // [B][N][ldx] the state (the handle's c)
  double* k[10];      // DOPRI5: k1..k6, k7 = f(y1) lands in k[1] (a_72 = e_2 = 0).  DOP853: k1..k10, k11 -> k[1], k12 -> k[2], f(ynew) -> k[3]
  double* y1;         // stage argument / (DOPRI5) new state
  double* ysti;       // DOPRI5: argument of stage 6 (stiffness detection); DOP853: the new state (K5 of dop853.f)
  double* d;          // [B][ODE_ND] per-lane reals
  int32_t* i;         // [B][ODE_NI] per-lane integers
  int32_t* counters;  // [64] lanes still inside the interval after a step (slot = step & 63)
};
hipError_t launch_ode_begin(const OdeArgs& a, hipStream_t stream);
hipError_t launch_ode_hinit(const OdeArgs& a, int half, hipStream_t stream);
hipError_t launch_ode_open(const OdeArgs& a, hipStream_t stream);
hipError_t launch_ode_stage(const OdeArgs& a, int stage, hipStream_t stream);     // stage 2..7
hipError_t launch_ode_control(const OdeArgs& a, hipStream_t stream);
// DOP853 (scipy 'dop853'): stage s = 2..12 of dop853.f, then control A (8th-order combination, new state, error estimate, accept /
// reject + controller), the right-hand side at the new state, control B (stiffness detection, commit, next step)
hipError_t launch_ode853_stage(const OdeArgs& a, int stage, hipStream_t stream);
int ode853_stage_dst(int stage);     // k buffer that receives f(stage argument)
hipError_t launch_ode853_control(const OdeArgs& a, int half, hipStream_t stream);

// ---- stabilised explicit integrator for stiff right-hand sides (pnp_rkc.hip; RKC of Sommeijer, Shampine, Verwer): the batched
// counterpart of the reference's odeint / ode('vode' | 'lsoda') drivers (calculator_old.py:946-963) ----
enum { RKC_T = 0, RKC_TEND, RKC_ABSH, RKC_H, RKC_HOLD, RKC_ERROLD, RKC_SPRAD, RKC_HMAX, RKC_SIGMA, RKC_DYNRM, RKC_W0, RKC_W1,
       RKC_BJM1, RKC_BJM2, RKC_ZJM1, RKC_ZJM2, RKC_DZJM1, RKC_DZJM2, RKC_D2ZJM1, RKC_D2ZJM2, RKC_ERR, RKC_ND = 24 };
enum { RKI_ACTIVE = 0, RKI_PHASE, RKI_J, RKI_M, RKI_LAST, RKI_NSTSIG, RKI_NEWSPC, RKI_JACATT, RKI_IDID, RKI_NSTEP_CALL, RKI_FIRST,
       RKI_HAVE_EV, RKI_STARTED, RKI_RHO_IT, RKI_INTERVAL, RKI_TOT_NSTEP, RKI_TOT_NACCPT, RKI_TOT_NREJCT, RKI_TOT_NFE, RKI_TOT_NFESIG,
       RKI_MAXM, RKC_NI = 24 };
struct RkcArgs {
  int32_t N, nx, ldx, nmax;
  int32_t mmax, slot, interval, pad_;
  int64_t B;
  double rtol, atol, max_step, dt;
  double* y;          // [B][N][ldx] y_n: the state (the handle's c)
  double* fn;         // f(y_n)
  double* F;          // f(arg) of the tick
  double* arg;        // where the right-hand side is wanted next: Y_{j-1} of the stage recurrence / power iterate / Euler probe
  double* yjm2;       // Y_{j-2}
  double* ev;         // eigenvector estimate of the power iteration (kept from one estimate to the next)
  double* d;          // [B][RKC_ND]
  int32_t* i;         // [B][RKC_NI]
  int32_t* counters;  // [64] lanes still inside the interval after a tick (slot = tick & 63)
};
hipError_t launch_rkc_begin(const RkcArgs& a, hipStream_t stream);
hipError_t launch_rkc_advance(const RkcArgs& a, hipStream_t stream);

}  // namespace pnp
