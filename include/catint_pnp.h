/*
 * catint_pnp.h -- C-ABI of the MI355X-native batched 1D Poisson-Nernst-Planck transport path.
 *
 * This is the drop-in boundary for the ONE hot path of sringe/CatINT that this repository
 * accelerates: the legacy finite-difference transport solve that sits under
 * `Calculator.run_single_step` (reference catint/calculator.py:408; legacy numeric seam
 * `Calculator.integrate_pnp(dx,nx,dt,nt,ntout,method)`, catint/calculator_old.py:210), batched
 * over B independent operating points ("lanes": one (phiM, bulk concentrations, wall fluxes)
 * tuple each -- the reference's descriptor loop calculator.py:204-212).
 *
 * Conventions
 *   - plain C, no HIP/torch types; every pointer is a HOST pointer to C-contiguous fp64/int32
 *     owned by the caller unless the name ends in _dev;
 *   - host-visible layouts are the reference's: a state is species-major flat `c[k*nx+i]`
 *     (calculator_old.py:508,562), batched as [B][N][nx];
 *   - every entry point returns 0 on success or a negative PNP_E* code and never calls exit();
 *     `pnp_last_error(h)` gives the message (replaces the reference's logger.error+sys.exit,
 *     e.g. calculator_old.py:109-111, :712-714);
 *   - the library owns the device buffers and one HIP stream behind the opaque handle; one
 *     handle per GPU; calls on one handle are not thread-safe, distinct handles are independent;
 *   - per-lane `status` replaces Comsol.check_error / the NaN test of calculator.py:409-414:
 *     0 ok, 1 Newton not converged (physical mode), 2 NaN/Inf in the state, 3 negative concentration.
 */
#ifndef CATINT_PNP_H
#define CATINT_PNP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNP_MAX_SPECIES 16
#define PNP_MAX_REACTIONS 16
#define PNP_MAX_REACTANTS 4

/* error codes */
#define PNP_OK 0
#define PNP_EINVAL (-1)      /* bad argument / unsupported configuration */
#define PNP_ENOMEM (-2)      /* device or host allocation failed */
#define PNP_EDEVICE (-3)     /* HIP runtime error (no GPU, launch failure, ...) */
#define PNP_ESTATE (-4)      /* call order violated (e.g. step before set_batch) */

/* integrator, mirrors the reference's calculator names (calculator_old.py:107, :1121-1140) */
#define PNP_METHOD_CRANK_NICOLSON 0   /* integrate_Crank_Nicolson, calculator_old.py:457-564 */
#define PNP_METHOD_FTCS 1             /* integrate_FTCS,           calculator_old.py:976-1029 */
/* physical mode: what the reference's production path asks COMSOL for (comsol_wrapper.py:145,158; model stated by
 * comsol_model.py) -- fully implicit coupled Poisson + Nernst-Planck, damped Newton per timestep
 * (or stationary), block-tridiagonal Jacobian solved by block cyclic reduction.  cfg.pb_mode must be PNP_PB_DD:
 * pb[b][0] = phiM of lane b, pb[b][1] = bulk potential; cfg.use_migration/lax_friedrich are ignored. */
#define PNP_METHOD_NEWTON 2
#define PNP_NEWTON_MAX_SPECIES 8
#define PNP_MAX_WALL_REACTIONS 8

/* Poisson boundary combination = which two slots of tp.pb_bound are set
 * (transport.py:1296-1311; branches of get_potential_and_gradient calculator_old.py:776-803) */
#define PNP_PB_DD 0            /* potential wall + potential bulk   (:780-786) */
#define PNP_PB_VWALL_GBULK 1   /* potential wall + gradient bulk    (reference default, transport.py:207-210) */
#define PNP_PB_GWALL_VBULK 2   /* gradient wall  + potential bulk */
#define PNP_PB_VWALL_GWALL 3   /* potential wall + gradient wall */
#define PNP_PB_VBULK_GBULK 4   /* potential bulk + gradient bulk */

/* lane status */
#define PNP_STATUS_OK 0
#define PNP_STATUS_MAXIT 1      /* physical mode: Newton did not reach tol within maxit iterations */
#define PNP_STATUS_NAN 2
#define PNP_STATUS_NEGATIVE 3

typedef struct pnp_handle pnp_handle;

/* Problem-wide configuration: what the reference keeps on `tp` and `Calculator` and that is
 * shared by every operating point of a sweep. */
typedef struct pnp_config {
  int32_t struct_size;      /* = sizeof(pnp_config), ABI guard */
  int32_t device;           /* HIP device ordinal */
  int32_t nspecies;         /* N  = tp.nspecies          (transport.py:264) */
  int32_t nx;               /* tp.nx, INCLUDING both boundary points (transport.py:459-460) */
  int32_t method;           /* PNP_METHOD_* */
  int32_t pb_mode;          /* PNP_PB_* */
  int32_t lax_friedrich;    /* '--LF' suffix (calculator_old.py:95-105) */
  int32_t use_migration;    /* tp.use_migration (transport.py:316-319) */
  int64_t batch_capacity;   /* max number of operating points held on this GPU */
  double dx;                /* tp.dx */
  double dt;                /* tp.dt */
  double beta;              /* tp.beta = 1/(R T)         (transport.py:312) */
  double eps;               /* tp.eps  = eps_r*eps_0     (transport.py:311) */
} pnp_config;

/* ---- lifetime ------------------------------------------------------------------------- */
int pnp_create(const pnp_config* cfg, pnp_handle** out);
void pnp_destroy(pnp_handle* h);
const char* pnp_last_error(const pnp_handle* h); /* h may be NULL: last create() error */
const char* pnp_version(void);

/* Debug / tuning switch of one handle: which kernel family runs, workspace sizes, probes.  key = the name of the environment variable
 * without its CATINT_ prefix (NEWTON_KERNEL = auto | generic | team | sweep | both | lane | lane2 | lane4 | workgroup, LANE_FUSED, LANE_RECORDS = f64 | f32 (the
 * columns T of the lane kernel's block-Thomas records in single precision: an inexact Newton update, the same converged state),
 * NEWTON_TEAM_THREADS, NEWTON_REGS, NEWTON_BLOCKS, NEWTON_LANE_GROUPS, NEWTON_SWEEP_BLOCKS, LANE_PIVOT_LIMIT, LANE_ORDER, PNP_KERNEL, PNP_WAVES_PER_GRID,
 * PNP_SPECIES_PER_WAVE, PNP_STEP_STREAMS, PNP_ALTERNATE_ROWS, PNP_ST_WAVES_PER_CU, PNP_NO_POST_UPLOAD_DISPATCH).  The environment is
 * read once, in pnp_create, as the defaults of the new handle; the library never reads it afterwards, so handles in one process
 * are configured independently.  PNP_EINVAL for an unknown key, PNP_ESTATE for an option that sizes a buffer already allocated.
 * Results do not depend on these switches beyond the documented rounding differences between kernel families. */
int pnp_set_option(pnp_handle* h, const char* key, const char* value);

/* ---- problem-wide parameters ------------------------------------------------------------ */
/* D[N] = tp.D (m^2/s, transport.py:423-434); charges[N] = tp.charges = z*F (transport.py:1271). */
int pnp_set_species(pnp_handle* h, const double* D, const double* charges);

/* Mass-action source terms of get_rates (calculator_old.py:159-208), FTCS only.
 * Reaction r: lhs[r*PNP_MAX_REACTANTS + j] (j < n_lhs[r]) and rhs[...] are species indices,
 * kf[r], kr[r] = tp.reactions[r]['rates']. Order matters (the reference's overwrite quirk). */
int pnp_set_reactions(pnp_handle* h, int32_t nreactions, const int32_t* n_lhs, const int32_t* lhs,
                      const int32_t* n_rhs, const int32_t* rhs, const double* kf, const double* kr);

/* ---- the batch of operating points ------------------------------------------------------- */
/* c0[B][N][nx]   initial concentrations = tp.c0 per lane (transport.py:1396-1412); its last grid
 *                point is also the bulk Dirichlet value C0[(k+1)*nx-1] (calculator_old.py:540)
 * pb[B][4]       {potential wall, potential bulk, gradient wall, gradient bulk} = tp.pb_bound
 *                (slots not selected by cfg.pb_mode are ignored)
 * vzeta[B]       tp.system['vzeta'] (calculator_old.py:529)
 * flux[B][N]     tp.flux_bound[k,0], wall flux (calculator_old.py:531, :1001)
 * For batches of 768 operating points or more the finite-difference handle also evaluates the potential and its gradient of the
 * uploaded state right away (two [capacity][row pitch] buffers, allocated on first use and counted in pnp_device_bytes; one more
 * kernel and synchronisation per upload): the dispatch keeps the first large launch after an upload at the sustained rate.
 */
int pnp_set_batch(pnp_handle* h, int64_t B, const double* c0, const double* pb, const double* vzeta,
                  const double* flux);

/* Update only the wall fluxes / wall potentials between SCF iterations (calculator.py:373-385). */
int pnp_set_flux(pnp_handle* h, const double* flux /* [B][N] */);
int pnp_set_pb(pnp_handle* h, const double* pb /* [B][4] */, const double* vzeta /* [B] */);

/* ---- the hot path -------------------------------------------------------------------------- */
/* Advance every lane by `nsteps` passes of the integrator's time-loop body. The state stays on
 * the device. `steps_per_launch` <= 0 lets the library choose (fused multi-step launches);
 * 1 forces one kernel launch per timestep with the full state read from and written to HBM.
 * A call of several launches over a large batch cuts the batch into row chunks whose launch sequences run on HIP streams of the
 * handle's own (operating points are independent); the call forks from and joins into the handle's stream, so callers see the
 * ordering of a single stream.  pnp_step_row_chunks tells how many chunks a call of `launches` launches uses. */
int pnp_step(pnp_handle* h, int32_t nsteps, int32_t steps_per_launch);
int32_t pnp_step_row_chunks(const pnp_handle* h, int32_t launches);

/* integrate_pnp (calculator_old.py:210): runs the reference's loop (n = 1..nt-1 for CN,
 * 0..nt-1 for FTCS) and copies the flattened state of every lane out at the steps listed in
 * itout[n_out] (tp.itout, calculator_old.py:140-152; ascending).
 * cout[n_out][B][N][nx]; status[B] (nullable). */
int pnp_integrate(pnp_handle* h, int32_t nt, const int32_t* itout, int32_t n_out, double* cout,
                  int32_t* status);

/* Method-of-lines right-hand side, ode_func (calculator_old.py:827-935), for the B lanes of the current batch
 * (their pb / flux / species / reactions): dcdt[B][N][nx] = f(c[B][N][nx]).  The reference hands this function
 * to scipy.integrate.odeint / ode ('odeint','lsoda','dopri5','dop853', calculator_old.py:946-963); the host side
 * (catint_amd/calculator.py) does the same with this entry point.  The state on the device is not touched. */
int pnp_mol_rhs(pnp_handle* h, const double* c, double* dcdt);

/* The reference's calc='dopri5' path with the integrator itself on the device: scipy.integrate.ode(ode_func).set_integrator(
 * 'dopri5', nsteps=10000), r.integrate(r.t + dt) once per interval (calculator_old.py:955-963).  scipy's 'dopri5' is Hairer's
 * DOPRI5 (Dormand-Prince 5(4), error norm sqrt(mean((e_i/(atol + rtol max(|y_i|,|ynew_i|)))^2)), Lund-stabilised controller,
 * stiffness detection); every lane runs it with its own step size, no host round trip per right-hand side.  Defaults (fields left
 * at 0) are scipy's: rtol 1e-6, atol 1e-12, nsteps 500, safety 0.9, ifactor 10, dfactor 0.2, beta 0 (-> 0.04), max_step 0 (-> the
 * interval), first_step 0 (-> HINIT), nstiff 1000. */
typedef struct pnp_ode_params {
  int32_t struct_size;   /* sizeof(pnp_ode_params) */
  int32_t nsteps;        /* NMAX: attempted steps per interval */
  double rtol, atol;
  double first_step, max_step;
  double safety, ifactor, dfactor, beta;
  int32_t nstiff;
  int32_t check_every;   /* steps enqueued between two reads of the "lanes left" counter (default 4) */
} pnp_ode_params;
/* Integrates the state on the device (pnp_set_batch) over nt intervals of cfg.dt starting at t = 0.  cout[n_out][B][N][nx]: the
 * state after interval n = itout[j] (0-based: the state at (n+1) dt, the indexing of the reference's `sol` list, :959-969);
 * idid[B]: DOPRI5's IDID of the lane's last call (1 ok, -2 nsteps exceeded, -3 step size too small, -4 stiff) -- a failed lane
 * stays at the state of its last accepted step, as r.successful() ends the reference's loop; stats[B][5] (nullable): attempted
 * steps, accepted, rejected, right-hand-side evaluations, interval of the last call; t_end[B] (nullable): time reached. */
int pnp_integrate_dopri5(pnp_handle* h, const pnp_ode_params* p, int32_t nt, const int32_t* itout, int32_t n_out, double* cout,
                         int32_t* idid, int64_t* stats, double* t_end);
/* The same for calc='dop853' (scipy's 'dop853' = Hairer's DOP853: 12 stages of order 8, error estimate from the embedded 5th- and
 * 3rd-order formulas): same arguments; the defaults are scipy's for this integrator (dfactor 0.3, ifactor 6, no Lund stabilisation).
 * scipy's build evaluates f(x, y) once more at the start of every step (same operands, same result); that evaluation is not done
 * and not counted here: stats[3] = 2 per interval + 11 per attempted step + 1 per accepted step. */
int pnp_integrate_dop853(pnp_handle* h, const pnp_ode_params* p, int32_t nt, const int32_t* itout, int32_t n_out, double* cout,
                         int32_t* idid, int64_t* stats, double* t_end);

/* The batched counterpart of the reference's STIFF drivers of the method of lines -- scipy.integrate.odeint (LSODA,
 * calculator_old.py:946-948) and ode('vode' | 'lsoda') (:955-963), which integrate one operating point per call on the host with
 * implicit multistep formulas.  On the device every lane runs RKC (Sommeijer, Shampine, Verwer 1998: second-order Runge-Kutta-
 * Chebyshev, m stages per step cover 0.65 m^2 / rho of the negative real axis; m follows from a spectral radius the integrator
 * estimates itself, the local error is controlled per lane with rtol / atol as in the other integrators): nothing but right-hand
 * sides, steps far beyond the explicit limit.  Results agree with odeint within the tolerance, not bit for bit (another formula).
 * Uses rtol, atol, nsteps (attempted steps per interval, default 100000), max_step, check_every (ticks between two reads of the
 * "lanes left" counter, default 16) of pnp_ode_params; the other fields are ignored.  Output as pnp_integrate_dopri5 (cout = state after
 * interval itout[j], i.e. at (n+1) dt).  idid[B]: 1 ok, -2 nsteps exceeded, -3 step size too small, -6 the power iteration for the
 * spectral radius did not converge; stats[B][7] (nullable): attempted, accepted, rejected steps, right-hand sides of the steps,
 * interval of the last call, right-hand sides of the spectral-radius estimates, largest stage count. */
int pnp_integrate_rkc(pnp_handle* h, const pnp_ode_params* p, int32_t nt, const int32_t* itout, int32_t n_out, double* cout,
                      int32_t* idid, int64_t* stats, double* t_end);

/* ---- physical mode (PNP_METHOD_NEWTON) -------------------------------------------------------------- */
typedef struct pnp_newton_params {
  int32_t struct_size;       /* = sizeof(pnp_newton_params) */
  int32_t wall_bc;           /* 0: phi(0) = phiM (Dirichlet);  1: Stern layer, eps dphi/dx = -C_S (phiM - phiPZC - phi(0))
                              *    (comsol_model.py:613, :982; tp.system['Stern capacitance'], ['phiPZC']) */
  int32_t maxit;             /* Newton iterations per solve (COMSOL maxiter 50, comsol_model.py:465-516) */
  int32_t error_estimate;    /* 0: converged when the scaled update < tol (default).  1: also when two consecutive undamped
                              *    iterations contract (upd_k < 0.1 upd_{k-1}) and the quadratic estimate upd_k^2/upd_{k-1} of the
                              *    error of the state just computed is < tol -- saves the iteration that only confirms convergence */
  double stern_capacitance;  /* F/m^2 */
  double phi_pzc;            /* V */
  double tol;                /* scaled update max(|dc|/(c + c_bulk), |dphi| beta max|q|) < tol on an undamped step.  One more exit
                              *    reports PNP_STATUS_OK: the ROUNDING FLOOR -- two consecutive undamped iterations whose scaled updates
                              *    are both < 100 tol and the second is not smaller than the first: the noise cond(J) eps of an
                              *    ill-conditioned Jacobian (stiff reactions), which no further iteration improves; the state is then
                              *    accurate to that noise (<= 100 tol), not to tol.  A linearly converging iteration (every update
                              *    smaller than the one before) does not qualify: it runs on to tol or to maxit (PNP_STATUS_MAXIT). */
  double dphi_max;           /* potential limiting per iteration (V); <= 0 disables */
  int32_t time_order;        /* pnp_step: 0 or 1 backward Euler (default); 2 = BDF2, the time stepping the reference asks COMSOL for
                              *    (comsol_model.py:518-531: tds time solver, "maxorder" 2): (3 c_n+1 - 4 c_n + c_n-1) / (2 dt), the first
                              *    step of a trajectory (after pnp_set_batch) backward Euler.  The lane kernels keep the history themselves (all timesteps of a
                              *    call in one launch); the workgroup-per-point kernels take one launch per timestep in this mode. */
  int32_t predictor;         /* pnp_step: 0 every timestep's Newton iteration starts from the previous state (default); 1 from the linear
                              *    extrapolation 2 u_n - u_n-1 of the two previous time levels (concentrations and potential), as a BDF
                              *    time stepper starts its corrector -- same equations and stopping rule, fewer iterations per step.
                              *    One launch per timestep in this mode; the first step of a trajectory starts from its initial state. */
} pnp_newton_params;
/* mpb_radius[N] (m, nullable = point ions): size-modified drift with phi0 = N_A sum a_k^3 c_k
 * (tp.species[sp]['MPB_radius'], comsol_model.py:1041-1063). */
int pnp_set_newton(pnp_handle* h, const pnp_newton_params* p, const double* mpb_radius);
/* Constant convection velocity v (m/s, along x) of the physical mode: the Nernst-Planck flux gains + c_i v, the reference's
 * tds.cdm1 "u" = tp.system['flow rate'] (comsol_model.py:901-903, :919; numbers only -- the reference also accepts a COMSOL
 * expression string there).  0 (default): none. */
int pnp_set_convection(pnp_handle* h, double velocity);
/* Non-uniform grid of the physical mode: x[nx] strictly increasing, x[0] = electrode, x[nx-1] = bulk boundary (the reference's
 * COMSOL mesh is refined towards the electrode: hmax = L/grid_factor_domain, lambda_D/grid_factor_bound at the boundaries,
 * comsol_model.py:588,593).  cfg.dx stays the reference length of the equation scaling (use x[1]-x[0]).  Default: x_i = i*dx. */
int pnp_set_grid(pnp_handle* h, const double* x);
/* First-order surface reactions solved implicitly with the transport (physical mode): reaction r contributes the flux
 * nu[r][k] * k[b][r] * c_{species[r]}(x=0) INTO the domain to species k of lane b (species[r] = -1: zeroth order), on top of
 * the prescribed pnp_set_flux values.  With rate constants k(phiM) evaluated per lane this is the fixed point the reference's
 * SCF loop (calculator.py:294-406: kinetics <-> transport with mixing) iterates towards when the kinetics are first order in
 * a surface concentration -- obtained in ONE solve and without host round trips.  n = 0 removes the table.
 * species[n], nu[n][N], k[B][n]; call after pnp_set_batch (k is per lane). */
int pnp_set_wall_kinetics(pnp_handle* h, int32_t n, const int32_t* species, const double* nu, const double* k);
/* Rate law of the table above beyond first order -- the forms the reference's user-defined flux equations take
 * (docs/source/topics/flux_definition.rst:90-160: rho*coverage*exp(-(Ga + alpha*F*(phiM - phi - phiEq))/RT), with a Langmuir
 * coverage K c/(1 + K c); handed to COMSOL as text by comsol_model.py:392-456):
 *   rate_r = k[b][r] * c_s/(1 + saturation[r] c_s) * exp(alpha[r] * (phiM[b] - phi(x=0)))
 * alpha[r] [1/V] (a cathodic Butler-Volmer branch has alpha = -alpha_BV F/RT; the equilibrium potential and activation energy
 * are constants inside k), saturation[r] [m^3/mol] >= 0.  phi(x=0) is the potential at the reaction plane, so with a Stern layer
 * (wall_bc = PNP_WALL_STERN) the driving force is the Stern-layer drop and the kinetics feed back on the double layer; both the
 * concentration and the potential derivative enter the Newton Jacobian.  In pnp_scf_cycle the factor is evaluated explicitly
 * from the surface potential of the previous transport solve.  n must equal the n of pnp_set_wall_kinetics, which resets both
 * arrays to zero (first order); either pointer may be NULL (= zeros). */
int pnp_set_wall_rate_law(pnp_handle* h, int32_t n, const double* alpha, const double* saturation);
/* Stationary solve of every lane from the current state as the initial guess (studies=['stat'], transport.py:811-812).
 * tol/maxit <= 0 keep the values of pnp_set_newton.  status[B] nullable. */
int pnp_solve_stationary(pnp_handle* h, double tol, int32_t maxit, int32_t* status);
/* One transport solve as the SCF loop needs it (calculator.py:373-400: new fluxes in, surface state out) in a single call with a
 * single synchronisation: optional pnp_set_flux(flux), then pnp_solve_stationary (nsteps = 0) or pnp_step(nsteps), then
 * pnp_get_surface + pnp_get_status.  Any output pointer may be NULL. */
int pnp_solve_surface(pnp_handle* h, const double* flux, int32_t nsteps, double* csurf, double* vsurf, double* esurf,
                      int32_t* status);
/* Kinetics <-> transport self-consistency loop of every lane ON THE DEVICE -- the reference's Calculator.run_scf_cycle
 * (catint/calculator.py:294-406) with the analytic kinetic model of pnp_set_wall_kinetics where CatMAP sat (explicitly:
 * flux_k = sum_r nu_rk K_r[lane] max(c_s(r)(x=0), 0) from the MIXED surface concentrations, not inside the Jacobian).
 * Per iteration and lane: mixing factor decay every 40 iterations (:319-323), mixing with the previous iterate and the
 * negative-concentration fallback (:328-344), surface pH (:346-359), new fluxes (:373), one warm-started stationary
 * transport solve of the lanes still active (:385, restart=True :523), current densities flux*nel*F/nprod/10 and the signed
 * relative change evaluate_accuracy (:260-283, :389-402); a lane leaves the loop when accuracy <= tau_scf and no surface
 * concentration is negative, a lane whose solve did not converge reports -1 (COMSOL's unreachable-flux answer).  No host
 * round trip per iteration: the host reads one counter every `check_every` iterations.
 * The caller runs the first iteration(s) (first transport solve, with whatever continuation it needs) and hands over the
 * loop state together with a CONVERGED transport state on the device: a lane whose later solve fails is put back to the
 * state of its last converged solve.  All arrays of pnp_scf_state are host arrays, read at entry, written back at exit. */
typedef struct pnp_scf_params {
  int32_t struct_size;   /* = sizeof(pnp_scf_params) */
  int32_t istep;         /* iterations already done (>= 1) */
  int32_t max_iter;      /* the loop ends after this iteration number at the latest */
  int32_t check_every;   /* host look-ups of the active-lane counter (1..32, <= 0: 8) */
  int32_t species_H;     /* index of H+ (surface pH = -log10(c/1000)), or -1 */
  int32_t species_OH;    /* index of OH- (used when there is no H+: 14 + log10(c/1000)), or -1 */
  double tau_scf;        /* convergence threshold on the relative current-density change */
  double faraday;        /* unit_F of the caller (catint/units.py) */
} pnp_scf_params;

typedef struct pnp_scf_state {
  double* surface_concentration;       /* [B][N] sc */
  double* surface_concentration_old;   /* [B][N] sc_old */
  double* flux;                        /* [B][N] wall fluxes of the last iteration */
  double* current_density_old;         /* [B][N] mA/cm^2 of the last iteration */
  double* mix;                         /* [B] mixing factor */
  double* accuracy;                    /* [B] */
  double* surface_pH;                  /* [B] */
  double* surface_potential;           /* [B] */
  double* surface_efield;              /* [B] */
  int32_t* step_to_check;              /* [B] iteration of the last mixing-factor decay */
  int32_t* active;                     /* [B] 1: still iterating */
  int32_t* failed;                     /* [B] 1: NaN in the transport solve (nan_in_surface, calculator.py:409-414) */
} pnp_scf_state;

int pnp_scf_cycle(pnp_handle* h, const pnp_scf_params* p, const double* nel /* [N] or NULL */, const double* nprod /* [N] or NULL */,
                  pnp_scf_state* s, int32_t* iterations /* last iteration number that had active lanes, nullable */);

/* Newton iterations each lane spent in the most recent pnp_step / pnp_solve_stationary call, summed over its
 * timesteps; a solve that hit maxit counts maxit+1. */
int pnp_get_newton_iterations(pnp_handle* h, int32_t* iters /* [B] */);
/* Overwrite the potential rows (initial guess of the physical mode), phi[B][nx]. */
int pnp_set_potential(pnp_handle* h, const double* phi);
/* Overwrite the state (concentrations c[n][N][nx], potential phi[n][nx]) of the n lanes lanes[0..n) and leave every other lane, the
 * iteration counters, the bulk values and the status flags alone: how a lane recovered elsewhere -- on a finer continuation ramp or
 * a finer mesh, the reference's rerun ladder (catint/calculator.py:466-531) -- comes back into the batch without the whole state
 * crossing PCIe.  phi may be NULL (potential rows unchanged). */
int pnp_set_lanes(pnp_handle* h, int64_t n, const int64_t* lanes, const double* c, const double* phi);
/* Restrict the following pnp_step / pnp_solve_stationary / pnp_solve_surface calls of the physical mode to the lanes with a non-zero
 * mask[b] (the others keep state, status and iteration counters); mask == NULL: all lanes again.  The mask is copied. */
int pnp_set_lane_mask(pnp_handle* h, const int32_t* mask /* [B] or NULL */);

/* Debug: the order in which the most recent lane-kernel launch of this handle dealt the operating points to its slots (slot s = group *
 * points per group + lane holds operating point perm[s]; most expected Newton iterations first, pnp_capi.hip: lane_order).  Returns the
 * number of slots filled (0: the last solve did not use an order -- another kernel family, fewer than 64 points, LANE_ORDER = 0);
 * perm[that many] (nullable). */
int64_t pnp_get_lane_order(pnp_handle* h, int32_t* perm);

/* Physical mode: choose the kernel family by MEASUREMENT on this device and this batch instead of by the library's thresholds (which
 * were measured on the devices of one pool; devices differ by ~10 %).  Every family that supports the shape -- pnp_autotune_name(i),
 * i < PNP_AUTOTUNE_CHOICES: lane4, lane2, lane, lane+fused, workgroup (the library's choice among the workgroup-per-point / team /
 * sweep kernels), team, sweep, both -- takes one warm-up timestep and then `nsteps` timed ones from the handle's current state; the
 * state, the BDF2 history, status and iteration counts are put back after every trial, so the call leaves the trajectory untouched.
 * The fastest family among those that solved the most operating points becomes the handle's NEWTON_KERNEL / LANE_FUSED option (as
 * pnp_set_option would set it: it stays until changed); the workspaces of the others are freed.  ms_per_step[PNP_AUTOTUNE_CHOICES]
 * (nullable) receives the measured time per timestep of every family, -1 where it does not apply; *chosen (nullable) the index.
 * Families agree to the Newton tolerance, not to the bit: tune once, before the steps whose results are compared.
 * Replaces nothing in the reference (COMSOL picks its own linear solver, comsol_model.py:465-516). */
/* Physical mode, lane kernels (large batches): put the workspace where it runs fastest.  On some devices the rate of an HBM-bound launch
 * takes one of a few discrete values that is decided by where its multi-GB workspace lies in device memory -- fixed for the lifetime
 * of the allocation, different between allocations (1.52 / 1.66 / 1.83e6 timesteps/s at 32 768 x 8 x 512).  The workspace is allocated up
 * to `trials` times (the earlier ones stay allocated meanwhile, so that each lands elsewhere; a trial that does not fit ends the
 * search), `nsteps` timesteps are timed on each from the handle's current state -- which is put back afterwards, history, status and
 * iteration counts included -- the fastest is kept and the others are freed.  ms_per_step[trials] (nullable): time per timestep of
 * every trial, -1 where none was made.  A batch no lane kernel takes: nothing happens.  No reference counterpart. */
int pnp_tune_placement(pnp_handle* h, int32_t nsteps, int32_t trials, double* ms_per_step);

#define PNP_AUTOTUNE_CHOICES 8
const char* pnp_autotune_name(int32_t i);
int32_t pnp_autotune_default(const pnp_handle* h);      /* index of the family the library's thresholds choose for the current batch; -1: no batch */
int pnp_autotune(pnp_handle* h, int32_t nsteps, double* ms_per_step, int32_t* chosen);

/* ---- read-back ------------------------------------------------------------------------------ */
/* Any pointer may be NULL. c[B][N][nx]; v, grad_v, lapl_v [B][nx] are the Poisson solve of the
 * most recent step = tp.potential, -tp.efield, -tp.total_charge/eps (calculator_old.py:816-818). */
int pnp_get_state(pnp_handle* h, double* c, double* v, double* grad_v, double* lapl_v);
/* Surface observables consumed by the SCF loop (comsol_reader.py:205-207, :278-279):
 * csurf[B][N] = c(x=0), vsurf[B] = v(x=0), esurf[B] = -grad_v(x=0). */
int pnp_get_surface(pnp_handle* h, double* csurf, double* vsurf, double* esurf);
int pnp_get_status(pnp_handle* h, int32_t* status /* [B] */);

/* ---- measurement hooks (bench.py; HIP events on the handle's own stream) ---------------------- */
int pnp_synchronize(pnp_handle* h);
int pnp_timer_start(pnp_handle* h);
int pnp_timer_stop(pnp_handle* h, float* elapsed_ms); /* synchronises the stream */
/* bytes of device memory held, and the device row pitch (in doubles) of one species row */
int64_t pnp_device_bytes(const pnp_handle* h);
int32_t pnp_row_pitch(const pnp_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* CATINT_PNP_H */
