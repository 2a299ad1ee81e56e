"""Host-side counterpart of the reference's ``Calculator`` for the finite-difference transport path,
batched over operating points and backed by the HIP library (no CPU fallback).

Reference interface mirrored here (paths relative to /root/reference):
  Calculator.__init__(transport, dt, tmax, ntout, calc, scale_pb_grid, tau_jacobi, tau_scf, mix_scf, mode)
                                      catint/calculator.py:55-143 == catint/calculator_old.py:54-157
  Calculator.integrate_pnp(dx,nx,dt,nt,ntout,method) -> list of flat [N*nx] arrays   calculator_old.py:210
  Calculator.run()  (descriptor sweep)                                              calculator.py:196-240
  Calculator.run_scf_cycle / evaluate_accuracy                                      calculator.py:260-406
The reference runs one descriptor point at a time; here every descriptor point is one lane of the
GPU batch (`integrate_pnp_batch`, `run`).
"""
import copy
import os

import numpy as np

from ._capi import PnpSolver, PnpError
from .host import pb_mode_from_bound
from .units import unit_F

CALC_LIST = ['FTCS', 'Crank-Nicolson', 'odeint', 'vode', 'lsoda', 'dopri5', 'dop853', 'odeint', 'odespy', 'comsol']
CALC_LIST = CALC_LIST + ['Newton']
GPU_CALCS = ('FTCS', 'Crank-Nicolson')
MOL_CALCS = ('odeint', 'lsoda', 'dopri5', 'dop853')   # method of lines: RHS on the GPU; 'dopri5' integrates on the device too, the others through scipy
# physical mode: what run_single_step asks COMSOL for (calculator.py:408-535, comsol_wrapper.py:145,158) solved on the
# GPU by the fully implicit coupled Newton kernel; 'comsol' is accepted as its name so reference scripts keep working
PHYSICAL_CALCS = ('comsol', 'Newton')
DEVICE_ODE_CALCS = ('dopri5', 'dop853')     # explicit Runge-Kutta integrators that run on the device (pnp_ode.hip)
# The reference's stiff driver (scipy's odeint = LSODA for calc='odeint' / 'lsoda': calculator_old.py:946-948) takes one operating point per call
# on the host.  Over a BATCH of operating points their role is played by the stiff integrator on the device (pnp_integrate_rkc:
# Runge-Kutta-Chebyshev with error control, pnp_rkc.hip) -- same tolerances, same output indexing, results within the tolerance.
DEVICE_STIFF_CALCS = ('odeint', 'lsoda')        # ('vode' raises a TypeError in the reference, SURVEY App. H: refused here as well)


class CalculatorError(ValueError):
    """Raised where the reference logs an error and calls sys.exit()."""


def make_itout(nt, ntout):
    """Output steps, calculator.py:126-138 == calculator_old.py:140-152."""
    itout = []
    for it in range(nt):
        if it == nt - 1:
            itout.append(it)
        elif it > 1 and it % int(nt / float(ntout)) == 0:
            itout.append(it)
    return itout


class Calculator(object):
    def __init__(self, transport=None, dt=None, tmax=None, ntout=1, calc=None, scale_pb_grid=None, tau_jacobi=1e-7,
                 tau_scf=5e-5, mix_scf=0.5, mode=None, desc_method='external', device=0):
        if transport is None:
            raise CalculatorError('No transport object provided for calculator.')
        self.tp = transport
        self.mode = mode
        self.tau_scf = tau_scf
        self.mix_scf = mix_scf
        self.device = device
        if calc is None:
            calc = self.tp.calc
        if calc is None:
            raise CalculatorError('No calculator name given (tp.set_calculator / calc=)')
        self.tp.ntout = ntout
        self.tp.desc_method = 'external'
        # '<name>[--LF|--<method>]', calculator.py:81-91
        self.use_lax_friedrich = False
        self.calc_method = None
        string = calc.split('--')
        if len(string) > 1:
            if string[-1] == 'LF':
                self.use_lax_friedrich = True
            else:
                self.calc_method = string[-1]
        self.calc = string[0]
        if self.calc not in CALC_LIST:
            raise CalculatorError('No calculator found with this name. Aborting.')
        if self.calc not in GPU_CALCS + MOL_CALCS + PHYSICAL_CALCS:
            raise CalculatorError("calculator '%s' is not part of the MI355X transport path "
                                  "(supported: %s)" % (self.calc, ', '.join(GPU_CALCS + MOL_CALCS + PHYSICAL_CALCS)))
        self.physical = self.calc in PHYSICAL_CALCS
        # roughness factor: every wall flux the COMSOL model prescribes is j_i = RF*flux_factor*flux_i (comsol_model.py:1000, :1134)
        self.RF = float(self.tp.system.get('RF', 1.0)) if self.physical else 1.0
        # the reference hands system['flow rate'] (a number or a COMSOL expression) to the convection velocity tds.cdm1 "u"
        # (comsol_model.py:901-903, :919): numbers are carried (pnp_set_convection: + c v in every flux), expressions are refused
        self.velocity = 0.0
        if self.physical and self.tp.system.get('flow rate') not in (None, 0, 0.0, '0', '0.0'):
            try:
                self.velocity = float(self.tp.system['flow rate'])
            except (TypeError, ValueError):
                raise CalculatorError("system['flow rate'] = %r: only a constant velocity (a number, m/s) is carried by the MI355X transport "
                                      "solver, not a COMSOL expression" % (self.tp.system['flow rate'],))
        if not self.physical and not getattr(self.tp, 'mesh_uniform', True):
            raise CalculatorError('the finite-difference integrators need a uniform mesh; graded meshes belong to calc="comsol"')
        if self.mode is None:      # COMSOL studies default to ['stat'] (transport.py:811-812); the FD integrators are transient
            self.mode = 'stationary' if self.physical else 'time-dependent'
        if scale_pb_grid is not None:
            raise CalculatorError('scale_pb_grid is broken in the reference (calculator_old.py:686-687) and not supported')
        self.scale_pb_grid = scale_pb_grid
        self.tau_jacobi = tau_jacobi
        # time mesh, calculator.py:105-116
        if dt is not None:
            self.tp.dt = dt
        if tmax is not None:
            self.tp.tmax = tmax
        if tmax is not None or dt is not None:
            self.tp.tmesh = np.arange(0, self.tp.tmax + self.tp.dt, self.tp.dt)
        else:
            self.tp.tmesh = np.arange(0, 1, 0.1)
            self.tp.dt = 0.1
        self.tp.nt = len(self.tp.tmesh)
        self.tp.itout = make_itout(self.tp.nt, self.tp.ntout)
        self.tp.ntout = len(self.tp.itout)

    # ------------------------------------------------------------------------------------------
    def _solver(self, B, pb_mode, dx, nx, dt):
        tp = self.tp
        # the method-of-lines calculators only use the handle for its RHS kernel; 'FTCS' carries the rate table
        s = PnpSolver(nspecies=tp.nspecies, nx=nx, dx=dx, dt=dt, beta=tp.beta, eps=tp.eps, D=tp.D, charges=tp.charges,
                      method=self.calc if self.calc in GPU_CALCS else 'FTCS', pb_mode=pb_mode,
                      lax_friedrich=self.use_lax_friedrich,
                      use_migration=tp.use_migration, batch_capacity=B, device=self.device)
        if getattr(tp, 'use_reactions', False) and getattr(tp, 'reactions', None):
            names = list(tp.species.keys())
            table = []
            for r in tp.reactions:
                rx = tp.reactions[r]
                if 'rates' not in rx:
                    continue
                table.append(([names.index(x) for x in rx['reactants'][0] if x in names],
                              [names.index(x) for x in rx['reactants'][1] if x in names],
                              float(rx['rates'][0]), float(rx['rates'][1])))
            if self.calc != 'Crank-Nicolson':
                s.set_reactions(table)
        elif self.calc == 'FTCS' and getattr(tp, 'reactions', None):
            # the reference's FTCS always calls get_rates (calculator_old.py:998)
            names = list(tp.species.keys())
            table = [([names.index(x) for x in rx['reactants'][0] if x in names],
                      [names.index(x) for x in rx['reactants'][1] if x in names],
                      float(rx['rates'][0]), float(rx['rates'][1]))
                     for rx in tp.reactions.values() if 'rates' in rx]
            if table:
                s.set_reactions(table)
        return s

    def integrate_pnp_batch(self, c0, pb, vzeta, flux, dx=None, nx=None, dt=None, nt=None, itout=None):
        """All lanes at once: c0 [B][N*nx], pb [B][4] (NaN = unset), vzeta [B], flux [B][N].
        Returns (cout [n_out][B][N*nx], status [B], (v, grad_v, lapl_v) of the last Poisson solve)."""
        tp = self.tp
        dx = tp.dx if dx is None else dx
        nx = tp.nx if nx is None else nx
        dt = tp.dt if dt is None else dt
        nt = tp.nt if nt is None else nt
        itout = tp.itout if itout is None else itout
        c0 = np.ascontiguousarray(c0, dtype=np.float64)
        B = c0.shape[0]
        pb = np.asarray(pb, dtype=np.float64).reshape(B, 4)
        modes = {pb_mode_from_bound(p) for p in pb}
        if len(modes) != 1:
            raise CalculatorError('all lanes of a batch must use the same pb_bound combination')
        with self._solver(B, modes.pop(), dx, nx, dt) as s:
            s.set_batch(c0, pb, vzeta, flux)
            if self.calc in DEVICE_STIFF_CALCS:
                # every lane its own adaptive RKC (pnp_integrate_rkc).  odeint returns the state at tmesh[n] (calculator_old.py:946-948:
                # row 0 is the initial state; the ode family's entry n is the state at (n + 1) dt, :959-969).
                shift = 1
                wanted = [int(n) for n in itout if n < nt]
                out = [n - shift for n in wanted if n - shift >= 0]
                opts = {k: v for k, v in getattr(self, 'ode_options', {}).items() if k in ('rtol', 'atol', 'nsteps', 'max_step', 'check_every')}
                # (intervals up to the last requested state only: odeint over tmesh covers nt - 1 of them, and a failure in an interval
                # nobody asked for must not flag the lane)
                cdev, idid, self.ode_stats, _ = s.integrate_rkc(max(out) + 1 if out else 1, out, **opts)
                cout = np.zeros((len(wanted), B, c0.shape[1]))
                for j, n in enumerate(wanted):
                    cout[j] = c0 if n - shift < 0 else cdev[out.index(n - shift)]
                self.ode_idid = idid
                status = (idid < 0).astype(np.int32)
            elif self.calc in DEVICE_ODE_CALCS:
                # every lane its own adaptive DOPRI5 / DOP853 (pnp_integrate_dopri5 / _dop853); output n = state at (n+1) dt (calculator_old.py:959-969);
                # status: 0 ok, 1 = the integrator gave up on the lane (nsteps / step size / stiffness: self.ode_idid)
                out = [int(n) for n in itout if n < nt]
                rk = s.integrate_dopri5 if self.calc == 'dopri5' else s.integrate_dop853
                cout, idid, self.ode_stats, _ = rk(nt, out, **dict({'nsteps': 10000}, **getattr(self, 'ode_options', {})))
                self.ode_idid = idid
                status = (idid < 0).astype(np.int32)
            else:
                cout, status = s.integrate(nt, itout)
            _, v, g, l = s.get_state()
        return cout, status, (v, g, l)

    def integrate_pnp(self, dx, nx, dt, nt, ntout, method):
        """Drop-in for the reference's single-operating-point seam (calculator_old.py:210): returns the list
        of flattened [N*nx] states at tp.itout and stores tp.potential / tp.efield / tp.total_charge
        (calculator_old.py:816-818)."""
        tp = self.tp
        if method != self.calc:
            raise CalculatorError('integrate_pnp: method %r differs from the calculator %r' % (method, self.calc))
        if self.calc in MOL_CALCS:
            return self._integrate_mol(dx, nx, dt, nt)
        cout, status, (v, g, l) = self.integrate_pnp_batch(
            tp.c0[None, :], tp.pb_array()[None, :], [tp.system['vzeta']], tp.flux_bound[None, :, 0],
            dx=dx, nx=nx, dt=dt, nt=nt, itout=tp.itout)
        tp.potential = v[0]
        tp.efield = -g[0]
        tp.total_charge = -l[0] * tp.eps
        self.status = int(status[0])
        return [cout[i, 0].copy() for i in range(cout.shape[0])]

    def _integrate_mol(self, dx, nx, dt, nt):
        """integrate_odeint (calculator_old.py:821-973): the method-of-lines right-hand side (ode_func :827-935) is the HIP kernel.
        calc='dopri5': the integrator runs on the device as well (pnp_integrate_dopri5: Hairer's DOPRI5 as scipy wraps it, one call
        per interval, :955-963) unless self.ode_on_device is False; 'odeint'/'lsoda'/'dop853': scipy's integrators drive the device
        right-hand side from the host.  Output indexing follows the reference: odeint/lsoda return the state at tmesh[n]; the `ode`
        family appends r.integrate(r.t+dt), so entry n is the state at (n+1)*dt (:959-963)."""
        import scipy.integrate as integrate
        tp = self.tp
        pb = tp.pb_array()[None, :]
        with self._solver(1, pb_mode_from_bound(pb[0]), dx, nx, dt) as s:
            s.set_batch(tp.c0[None, :], pb, [tp.system['vzeta']], tp.flux_bound[None, :, 0])
            if self.calc in DEVICE_ODE_CALCS and getattr(self, 'ode_on_device', True):
                out = [n for n in range(nt) if n in tp.itout]
                rk = s.integrate_dopri5 if self.calc == 'dopri5' else s.integrate_dop853
                cout, idid, stats, t_end = rk(nt, out, **dict({'nsteps': 10000}, **getattr(self, 'ode_options', {})))
                self.status = 0
                self.ode_idid, self.ode_stats = int(idid[0]), stats[0].copy()
                # a failed call ends the reference's loop after appending its result (:959-963): sol stops at that interval
                nsol = nt if idid[0] >= 0 else int(stats[0, 4]) + 1
                return [cout[j, 0].copy() for j, n in enumerate(out) if n < nsol]

            def f_ty(t, c):
                return s.mol_rhs(c[None, :])[0]

            if self.calc in ('lsoda', 'odeint'):     # :946-948
                sol = integrate.odeint(lambda c, t: f_ty(t, c), tp.c0, tp.tmesh, ml=tp.nspecies, mu=tp.nspecies)
            else:                                    # :955-963 (the reference's nsteps=10000 branch)
                r = integrate.ode(f_ty).set_integrator(self.calc, **dict({'nsteps': 10000}, **getattr(self, 'ode_options', {})))
                r.set_initial_value(tp.c0)
                sol = []
                while r.successful() and r.t < nt * dt:
                    sol.append(r.integrate(r.t + dt))
                sol = np.array(sol)
            c, v, g, l = s.get_state()
        self.status = 0
        return [np.array(sol[n, :]) for n in range(0, nt) if n in tp.itout and n < len(sol)]

    # -- physical mode ---------------------------------------------------------------------------
    def _physical_solver(self, B, nx=None, dx=None, dt=None, xmesh=None):
        """Handle of the implicit coupled solver with the boundary model of the COMSOL generator: Stern Robin wall
        (tp.system['Stern capacitance'] in muF/cm^2, ['phiPZC']; comsol_model.py:613,:982,:1167), bulk potential 0 V
        (:662-663), size-modified drift for species carrying 'MPB_radius' (:1041-1063).
        tp.system['wall potential'] = 'dirichlet' switches the Stern layer off (phi(0) = phiM)."""
        tp = self.tp
        nk = getattr(tp, 'newton', {})
        if xmesh is not None:
            xmesh = np.asarray(xmesh, float)
            nx, dx = len(xmesh), float(xmesh[1] - xmesh[0])
        s = PnpSolver(nspecies=tp.nspecies, nx=tp.nx if nx is None else nx, dx=tp.dx if dx is None else dx,
                      dt=tp.dt if dt is None else dt, beta=tp.beta, eps=tp.eps, D=tp.D, charges=tp.charges, method='Newton',
                      pb_mode=0, batch_capacity=B, device=self.device)
        radii = [float(tp.species[sp].get('MPB_radius', 0.0)) for sp in tp.species]
        cs = float(tp.system.get('Stern capacitance', 0.0)) * 1e-2          # muF/cm^2 -> F/m^2
        stern = tp.system.get('wall potential', 'stern') == 'stern' and cs > 0
        s.set_newton(wall_bc='stern' if stern else 'dirichlet', stern_capacitance=cs if stern else 0.0,
                     phi_pzc=float(tp.system.get('phiPZC', 0.0)), tol=nk.get('tol', 1e-8), maxit=nk.get('maxit', 50),
                     dphi_max=nk.get('dphi_max', 0.05), mpb_radius=radii if any(radii) else None,
                     # time-dependent mode: tp.newton['time_order'] = 2 steps with BDF2, the formula the reference's transient study
                     # asks COMSOL for (comsol_model.py:518-531: BDF, maxorder 2); default backward Euler
                     time_order=int(nk.get('time_order', 1)))
        if nk.get('records') in ('f32', 'f64'):          # tp.newton['records'] = 'f32': the lane kernel's record columns in single precision
            s.set_option('LANE_RECORDS', nk['records'])
        if xmesh is not None:
            s.set_grid(xmesh)
        elif not getattr(tp, 'mesh_uniform', True):
            s.set_grid(tp.xmesh)
        if getattr(self, 'velocity', 0.0):
            s.set_convection(self.velocity)
        if getattr(tp, 'use_reactions', False) and getattr(tp, 'reactions', None):      # tp.reactions[r]['reactants'], ['rates']
            names = list(tp.species.keys())
            s.set_reactions([([names.index(x) for x in rx['reactants'][0] if x in names],
                              [names.index(x) for x in rx['reactants'][1] if x in names],
                              float(rx['rates'][0]), float(rx['rates'][1]))
                             for rx in tp.reactions.values() if 'rates' in rx])
        return s

    def set_surface_kinetics(self, reactions):
        """Electrode kinetics solved WITH the transport instead of around it (physical mode only):
        reactions = [{'species': name (or None: zeroth order), 'rate': K(phiM[B]) -> [B] in m/s (mol m^-2 s^-1 if zeroth order),
        'stoichiometry': {species name: nu}, 'alpha': a (1/V, optional), 'saturation': K_sat (m^3/mol, optional)}], flux into the
        electrolyte nu*K*g(c_species(x=0))*exp(a*(phiM - phi(x=0))), g(c) = c/(1 + K_sat c) (educts negative,
        calculator.py:415-432).  alpha / saturation give the Butler-Volmer and Langmuir forms of the reference's user-defined flux
        equations (docs/source/topics/flux_definition.rst:90-160); without them the reaction is first order.  One stationary
        solve then returns what run_scf_cycle iterates towards."""
        if not self.physical:
            raise CalculatorError('surface kinetics are part of the physical mode (calc="comsol")')
        for r in reactions:
            if float(r.get('saturation', 0.0)) < 0.0:
                raise CalculatorError('surface kinetics: saturation must be >= 0')
            if float(r.get('saturation', 0.0)) != 0.0 and r.get('species') is None:
                raise CalculatorError('surface kinetics: saturation needs a species')
        self.surface_kinetics = list(reactions)

    def _apply_surface_kinetics(self, solver, phiM, lanes=None):
        rx = getattr(self, 'surface_kinetics', None)
        if not rx:
            return
        names = list(self.tp.species.keys())
        species = [names.index(r['species']) if r.get('species') is not None else -1 for r in rx]
        nu = np.zeros((len(rx), len(names)))
        for i, r in enumerate(rx):
            for sp, v in r['stoichiometry'].items():
                nu[i, names.index(sp)] = v
        def per_lane(rate):
            # a rate given per lane of the FULL batch (array, or a callable closing over such arrays) is cut down to the lanes of a
            # sub-batch -- the retry ladder solves failed lanes as a batch of their own
            v = np.asarray(rate(phiM) if callable(rate) else rate, float)
            if lanes is not None and v.ndim == 1 and v.shape[0] != phiM.shape[0] and v.shape[0] > int(np.max(lanes)):
                v = v[np.asarray(lanes)]
            return np.broadcast_to(v, phiM.shape)
        k = np.stack([per_lane(r['rate']) for r in rx], axis=1)
        alpha = [float(r.get('alpha', 0.0)) for r in rx]
        sat = [float(r.get('saturation', 0.0)) for r in rx]
        law = any(alpha) or any(sat)
        solver.set_wall_kinetics(species, nu, self.RF * k, alpha if law else None, sat if law else None)

    def surface_kinetic_fluxes(self, csurf, phiM, clip=False, reactions=None, vsurf=None):
        """Flux [B][N] into the electrolyte implied by set_surface_kinetics at the surface state csurf [B][N], vsurf [B] = phi(x=0)
        (needed when a reaction has an 'alpha'; clip: negative surface concentrations count as zero, as in the explicit SCF loop)."""
        names = list(self.tp.species.keys())
        out = np.zeros_like(csurf)
        for r in (reactions if reactions is not None else getattr(self, 'surface_kinetics', None)) or []:
            K = np.broadcast_to(np.asarray(r['rate'](phiM) if callable(r['rate']) else r['rate'], float), phiM.shape)
            cs = csurf[:, names.index(r['species'])] if r.get('species') is not None else 1.0
            if clip and r.get('species') is not None:
                cs = np.maximum(cs, 0.0)
            g = cs
            if float(r.get('saturation', 0.0)) != 0.0:
                g = g * (1.0 / (1.0 + float(r['saturation']) * cs))
            if float(r.get('alpha', 0.0)) != 0.0:
                if vsurf is None:
                    raise CalculatorError('surface_kinetic_fluxes: a reaction with alpha needs the surface potential')
                g = g * np.exp(float(r['alpha']) * (phiM - np.asarray(vsurf, float)))
            for sp, v in r['stoichiometry'].items():
                out[:, names.index(sp)] += v * K * g
        return out

    def solve_physical(self, solver, c0, phiM, flux, nramp=4, warm=False, retry=True):
        """One transport solve of every lane (run_single_step, calculator.py:408-535).
        Stationary mode: Newton from the current state (warm) or from the bulk state.  If lanes do not converge from the
        bulk state -- or the potentials are far from phiPZC, or surface kinetics are coupled in, where that is the rule --
        every lane walks a continuation path instead: the wall potential goes from phiPZC (uncharged interface) to its phiM
        in stages of at most tp.newton['dphi_stage'] (0.4 V since round 4, at least nramp = 4 stages; round 3: 0.2 V and 8.  The Newton iteration limits its own
        potential step, so wide stages cost iterations, not convergence: on the CO2R example 0.2 / 0.3 / 0.4 / 0.5 / 1.0 V = 11 / 8 / 6 / 5 / 3 stages take
        0.126 / 0.095 / 0.081 / 0.068 / 0.051 s, all 4096 lanes converge to the same answer (1e-9; tools/probe/co2r_stage_width.py ->
        profiles/r04_co2r_stage_width.jsonl); lanes that fail walk the path again with 2, 4, 8 x the stages, see below), the prescribed fluxes grow proportionally and the surface
        rate constants are evaluated at the stage potential, each stage warm-started from the previous one (the reference's
        parametric sweeps: flux_factor / PZC / CS ramps, transport.py:877-893, comsol_model.py:1147-1167).
        Time-dependent mode: tp.nt-1 backward-Euler steps.  retry=False: no rerun ladder for lanes that fail (the first solve of an
        SCF cycle, where a lane without a solution is an expected answer).  Returns status [B]."""
        phiM = np.asarray(phiM, float)
        B = len(phiM)
        pb = np.zeros((B, 4)); pb[:, 0] = phiM
        vz = np.zeros(B)
        if warm:
            solver.set_flux(self.RF * np.asarray(flux, float))
            if self.mode != 'stationary':
                solver.step(self.tp.nt - 1)
                return solver.get_status()
            return solver.solve_stationary()
        flux = self.RF * np.asarray(flux, float)        # what reaches the wall: j = RF * flux (comsol_model.py:1134)
        if self.mode != 'stationary':
            solver.set_batch(c0, pb, vz, flux)
            self._apply_surface_kinetics(solver, phiM)
            nk_ = getattr(self.tp, 'newton', {})
            # (a long run of a large batch: the lane kernels' workspace goes where it runs fastest first -- pnp_tune_placement, up to six
            # placements on two timesteps each, the state is put back; tp.newton['tune_placement'] forces it on or off)
            if nk_.get('tune_placement', B >= 16384 and self.tp.nt - 1 >= 40) and hasattr(solver, 'tune_placement'):
                solver.tune_placement(2, 6)
            solver.step(self.tp.nt - 1)
            return solver.get_status()
        nk = getattr(self.tp, 'newton', {})
        if nramp > 1 and 'min_stages' in nk:      # (tp.newton['min_stages']: the least number of stages of a continuation path)
            nramp = int(nk['min_stages'])
        stern = self.tp.system.get('wall potential', 'stern') == 'stern' and float(self.tp.system.get('Stern capacitance', 0.0)) > 0
        start = float(self.tp.system.get('phiPZC', 0.0)) if stern else 0.0
        span = float(np.abs(phiM - start).max())
        direct = nramp <= 1 or (span <= nk.get('direct_span', 0.6) and not getattr(self, 'surface_kinetics', None))
        if direct:
            solver.set_batch(c0, pb, vz, flux)
            self._apply_surface_kinetics(solver, phiM)
            st = solver.solve_stationary()
            if not (st != 0).any() or nramp <= 1:
                return st
        nst = max(int(nramp), int(np.ceil(span / nk.get('dphi_stage', 0.4))))
        self.continuation_stages = nst
        st = self._continuation(solver, c0, pb, vz, flux, phiM, start, nst)
        # Lanes that still fail: the reference reruns COMSOL up to 25 times with a load / non-linearity ramp half as coarse each time
        # (and a finer boundary mesh, which a handle cannot change: refine with tp.set_graded_mesh) -- calculator.py:455-531.  Here
        # only the failed lanes walk the path again, as a batch of their own, with 2, 4, 8 x the stages; what converges is patched
        # into the state of the main handle.
        self.retry_log = []
        rungs = int(nk.get('retry_rungs', 3)) if retry else 0
        mesh_rungs = int(nk.get('retry_mesh_rungs', 1)) if retry else 0
        for rung in range(1, rungs + mesh_rungs + 1):
            bad = np.flatnonzero(st != 0)
            if len(bad) == 0:
                break
            mesh = rung > rungs          # the last rungs refine the boundary mesh as well (grid_factor_bound *= 1.5, calculator.py:466-531)
            stages = nst * 2 ** min(rung, max(rungs, 1))
            xf = None
            if mesh:
                from .host import graded_mesh
                x = np.asarray(self.tp.xmesh, float)
                xf = graded_mesh(x[-1], (x[1] - x[0]) / 1.5 ** (rung - rungs), len(x))
            with self._physical_solver(len(bad), xmesh=xf) as sub:
                st_sub = self._continuation(sub, np.asarray(c0)[bad], pb[bad], vz[bad], np.asarray(flux)[bad], phiM[bad], start, stages,
                                            lanes=bad)
                c_sub, phi_sub = sub.get_state()[:2]
            good = st_sub == 0
            entry = {'rung': rung, 'stages': stages, 'lanes': bad.tolist(), 'recovered': [], 'mesh_refined': bool(mesh)}
            self.retry_log.append(entry)
            if good.any():
                c_sub = np.asarray(c_sub, float).reshape(len(bad), self.tp.nspecies, -1)[good]
                phi_sub = np.asarray(phi_sub, float).reshape(len(bad), -1)[good]
                if mesh:      # back onto the batch's mesh: the finer solution is the initial guess of the confirming solve there
                    x = np.asarray(self.tp.xmesh, float)
                    c_sub = np.stack([[np.interp(x, xf, row) for row in lane] for lane in c_sub])
                    phi_sub = np.stack([np.interp(x, xf, row) for row in phi_sub])
                # only the recovered lanes travel (pnp_set_lanes); counters, flags and the other lanes stay as they are ...
                solver.set_lanes(bad[good], c_sub, phi_sub)
                # ... and the main handle's own verdict on them is what is reported: one solve restricted to these lanes
                mask = np.zeros(B, np.int32)
                mask[bad[good]] = 1
                solver.set_lane_mask(mask)
                try:
                    st2 = solver.solve_stationary()
                finally:
                    solver.set_lane_mask(None)
                st = st.copy()
                st[bad[good]] = st2[bad[good]]
                entry['recovered'] = [int(b_) for b_ in bad[good] if st[b_] == 0]
        return st

    def _continuation(self, solver, c0, pb, vz, flux, phiM, start, nst, lanes=None):
        """nst stages from the uncharged interface to the operating point (wall potential, prescribed fluxes, rate constants), each
        warm-started from the previous one; lanes: indices of these lanes in the full batch (per-lane rate constants)."""
        st = None
        # (tp.newton['stage_tol'], default 1e-3: a looser tolerance for the stages BEFORE the operating point -- they only have to deliver
        # a start the last stage's Newton converges from; the last stage is solved to tp.newton['tol'] as always.  On the CO2R example
        # (4096 voltages, 11 stages): 277 k -> 228 k Newton iterations, 0.145 -> 0.126 s, surface concentrations and currents equal to
        # 1e-12 (tools/probe/co2r_stage_tol.py -> profiles/r04_co2r_stage_tol.jsonl).  0 / None: every stage to the full tolerance.)
        stage_tol = float(getattr(self.tp, 'newton', {}).get('stage_tol', 1e-3) or 0.0)
        for j in range(1, nst + 1):
            w = j / float(nst)
            pbj = pb.copy(); pbj[:, 0] = start + (phiM - start) * w
            if j == 1:
                solver.set_batch(c0, pbj, vz, flux * w)
            else:
                solver.set_pb(pbj, vz)
                solver.set_flux(flux * w)
            self._apply_surface_kinetics(solver, pbj[:, 0], lanes=lanes)
            st = solver.solve_stationary(stage_tol, 0) if (stage_tol > 0.0 and j < nst) else solver.solve_stationary()
            it = solver.newton_iterations()
            # (what bench.py reports next to the wall time: iterations spent by all lanes, and by the slowest lane of each stage -- a
            # lane kernel's launch lasts as long as its slowest operating point)
            self.newton_iterations_total = getattr(self, 'newton_iterations_total', 0) + int(it.sum())
            self.newton_iterations_slowest = getattr(self, 'newton_iterations_slowest', 0) + int(it.max())
        return st

    # ------------------------------------------------------------------------------------------
    def run_single_step(self, label=''):
        """One transport solve at the CURRENT operating point tp.system / tp.species[*]['flux'] (reference
        Calculator.run_single_step, calculator.py:408-535: comsol.run + check_error + reader) -- a batch of one lane.  The
        results go where the reference's reader puts them with update_last (comsol_reader.py:196-279): tp.species[sp]
        ['concentration','surface_concentration', ...] and tp.system['potential','efield','surface_potential','surface_pH', ...].
        Returns True when the solve converged (the reference's check_error + NaN test, :409-414, :457)."""
        tp = self.tp
        saved = (tp.descriptors, tp.alldata_names, tp.alldata)
        keys = list(tp.descriptors.keys())
        try:
            import collections
            tp.descriptors = collections.OrderedDict((k, [tp.system[k]]) for k in keys)
            tp.alldata_names = [[tp.system[keys[0]], tp.system[keys[1]]]]
            tp.alldata = [{'species': {}, 'system': {}}]
            # symbolic fluxes ('catmap', equations: transport.py:938-947) carry no number of their own -- the kinetics callback /
            # the implicit wall kinetics supply it; the prescribed part is zero
            if self.flux_sign_error():        # calculator.py:435-442
                self.change_flux_sign()
                self.flux_signs_changed = True
            tp.flux_bound[:, 0] = [0.0 if isinstance(tp.species[sp].get('flux', 0.0), str) else float(tp.species[sp].get('flux', 0.0))
                                   for sp in tp.species]
            self.run()
            d = tp.alldata[0]
        finally:
            tp.descriptors, tp.alldata_names, tp.alldata = saved
        for sp in tp.species:
            tp.species[sp].update(d['species'][sp])
        tp.system.update({k: v for k, v in d['system'].items() if k not in keys and k != 'status'})
        self.last_label = label
        return int(np.atleast_1d(self.status)[0]) == 0

    def flux_sign_error(self):
        """err_in_flux of the reference's run_single_step (calculator.py:415-432): an educt of an electrode reaction with a flux that
        is not negative, or a product with a positive flux whose sign is not +1 (never true -- kept as written).  Species in
        system['exclude species'] other than H+ / OH- are skipped; symbolic fluxes (strings) count as zero."""
        tp = self.tp
        excl = tp.system.get('exclude species', [])

        def num(sp):
            f = tp.species[sp].get('flux', 0.0)
            return 0.0 if isinstance(f, str) else float(f)
        error = False
        for name in getattr(tp, 'electrode_reactions', {}) or {}:
            educts, products = tp.electrode_reactions[name]['reaction'][0], tp.electrode_reactions[name]['reaction'][1]
            for educt in educts:
                if (educt in excl and educt not in ['H+', 'OH-']) or educt not in tp.species:
                    continue
                if num(educt) != 0.0 and num(educt) / abs(num(educt)) != -1:
                    error = True
            for prod in products:
                if (prod in excl and prod not in ['H+', 'OH-']) or prod not in tp.species:
                    continue
                if num(prod) > 0.0 and num(prod) / abs(num(prod)) != 1:
                    error = True
        return error

    def change_flux_sign(self):
        """change_flux_sign (calculator.py:433-446): every educt and product of the electrode reactions (outside the excluded species)
        gets the opposite flux, each species once."""
        tp = self.tp
        excl = tp.system.get('exclude species', [])
        done = []
        for name in getattr(tp, 'electrode_reactions', {}) or {}:
            for side in (0, 1):
                for sp in tp.electrode_reactions[name]['reaction'][side]:
                    if sp in excl or sp in done or sp not in tp.species:
                        continue
                    f = tp.species[sp].get('flux', 0.0)
                    if not isinstance(f, str):
                        tp.species[sp]['flux'] = -1.0 * f
                    done.append(sp)

    # ------------------------------------------------------------------------------------------
    def run(self):
        """Descriptor sweep (calculator.py:196-240) with every descriptor point as one GPU lane.
        Fills tp.alldata[i]['species'|'system'] with the field contract of comsol_reader.py:196-326
        that the FD path can provide (concentration, surface_concentration, potential, efield,
        surface_potential, surface_efield, charge_density)."""
        tp = self.tp
        keys = list(tp.descriptors.keys())
        lanes = tp.alldata_names
        B = len(lanes)
        pb = np.zeros((B, 4)); vz = np.zeros(B)
        for i, (v1, v2) in enumerate(lanes):
            system = dict(tp.system)
            system[keys[0]] = v1
            system[keys[1]] = v2
            if 'phiM' in keys and 'vzeta' not in (keys[0], keys[1]) and getattr(tp, 'vzeta_follows_phiM', True):
                system['vzeta'] = system['phiM']
            pb[i] = tp.pb_array(system)
            vz[i] = system['vzeta']
        c0 = np.repeat(tp.c0[None, :], B, axis=0)
        flux = np.repeat(tp.flux_bound[None, :, 0], B, axis=0)
        if self.physical:
            phiM = np.array([dict(tp.system, **{keys[0]: v1, keys[1]: v2})['phiM'] for (v1, v2) in lanes], float)
            import time as _time
            with self._physical_solver(B) as s:
                _t0 = _time.perf_counter()
                status = self.solve_physical(s, c0, phiM, flux)
                self.solve_seconds = _time.perf_counter() - _t0        # transport solves only (incl. their host<->device traffic)
                cfin, v, g, l = s.get_state()
                self.newton_iterations = s.newton_iterations()
                if getattr(self, 'surface_kinetics', None):
                    self.kinetic_flux = self.surface_kinetic_fluxes(cfin[:, :, 0], phiM, vsurf=v[:, 0])
            cout = cfin.reshape(1, B, tp.nspecies * tp.nx)
        else:
            cout, status, (v, g, l) = self.integrate_pnp_batch(c0, pb, vz, flux)
        self.status = status
        kf = getattr(self, 'kinetic_flux', None) if self.physical else None
        self.fill_alldata_batch(cout[-1].reshape(B, tp.nspecies, tp.nx), v, g, l, flux, status, kf)
        return cout

    # ------------------------------------------------------------------------------------------
    def fill_alldata(self, i, cfin, v, g, l, flux, status, kinetic_flux=None):
        """Descriptor point i alone (fill_alldata_batch with one lane)."""
        self.fill_alldata_batch(np.asarray(cfin)[None], np.asarray(v)[None], np.asarray(g)[None], np.asarray(l)[None],
                                np.asarray(flux)[None], [status], None if kinetic_flux is None else np.asarray(kinetic_flux)[None], first=i)
        return self.tp.alldata[i]

    def fill_alldata_batch(self, cfin, v, g, l, flux, status, kinetic_flux=None, first=0):
        """Descriptor points first ... first + B - 1: tp.alldata[i]['species'|'system'] in the field contract of the reference's reader
        (comsol_reader.py:196-326) from the arrays a transport solve leaves behind -- cfin [B][N][nx], potential v,
        gradient g and charge row l [B][nx], prescribed wall fluxes flux [B][N].  Host arithmetic only: the derived arrays are computed
        64 lanes at a time (cache-sized pieces: the ~20 temporaries of 64 lanes x N x nx stay in L2; pure numpy, so the pieces of a large
        sweep run on a pool of threads), then the dictionaries are filled; the per-point entries are rows of the pieces' arrays.  (A sweep
        of 4096 voltages spent 0.5 s here point by point, three times its transport solves.  tests/test_results_io.py walks the
        reference's plotting accesses over the output without a GPU.)"""
        B = len(cfin)
        pieces = [(a, min(a + 64, B)) for a in range(0, B, 64)]

        def derive(ab):
            a, e = ab
            return self._alldata_arrays(cfin[a:e], v[a:e], g[a:e], l[a:e], flux[a:e], None if kinetic_flux is None else kinetic_flux[a:e])
        if len(pieces) > 2:
            import concurrent.futures
            with concurrent.futures.ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1, len(pieces))) as pool:
                derived = list(pool.map(derive, pieces))
        else:
            derived = [derive(ab) for ab in pieces]
        for (a, e), arrays in zip(pieces, derived):
            self._alldata_fill(first + a, arrays, status[a:e])

    def _alldata_arrays(self, cfin, v, g, l, flux, kinetic_flux):
        """The arrays behind the dictionaries of a piece of the sweep (no shared state: runs on any thread)."""
        tp = self.tp
        from .units import unit_NA
        cfin = np.array(cfin, dtype=float)                      # (own copies: the dictionaries keep rows of these arrays)
        v = np.array(v, dtype=float)
        g = np.asarray(g, float)
        l = np.asarray(l, float)
        flux = np.asarray(flux, float)
        B = cfin.shape[0]
        out = {'cfin': cfin, 'v': v, 'efield': -g, 'rho': -l * tp.eps}
        if not self.physical:
            return out
        names = list(tp.species.keys())
        radii = np.array([float(tp.species[sp].get('MPB_radius', 0.0)) for sp in names])
        # derived fields of the COMSOL reader (comsol_reader.py:57-90, :196-230, :241-261)
        gamma = 1.0 / (1.0 - (unit_NA * radii[None, :, None] ** 3 * cfin).sum(axis=1))                 # [B][nx]
        jwall = flux + (np.asarray(kinetic_flux, float) if kinetic_flux is not None else 0.0)          # [B][N]
        with np.errstate(divide='ignore', invalid='ignore'):
            if 'H+' in names:
                ph = -np.log10(cfin[:, names.index('H+')] / 1000.)
            elif 'OH-' in names:
                ph = 14 + np.log10(cfin[:, names.index('OH-')] / 1000.)
            else:
                ph = None
            pH = None if ph is None else ph - np.log10(gamma)
        # derived electrolyte quantities of the COMSOL model (comsol_model.py:1010-1040), on the cell edges
        x = np.asarray(tp.xmesh, float)
        h = np.diff(x)
        z = tp.charges / unit_F
        cmid = 0.5 * (cfin[:, :, 1:] + cfin[:, :, :-1])
        um = tp.D * tp.beta                                            # mobility D/(RT)
        kappa = unit_F ** 2 * ((z ** 2 * um)[None, :, None] * cmid).sum(axis=1)        # rho_c, S/m
        with np.errstate(divide='ignore', invalid='ignore'):
            w = -np.log(1.0 / gamma)                                                   # -ln(1-phi0)
        u = (tp.charges * tp.beta)[None, :, None] * np.diff(v, axis=1)[:, None, :] + np.diff(w, axis=1)[:, None, :]
        with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
            Bu = np.where(np.abs(u) < 1e-8, 1.0 - 0.5 * u, u / np.expm1(u))
            Dh = tp.D[None, :, None] / h[None, None, :]
            jtot = -Dh * ((Bu + u) * cfin[:, :, 1:] - Bu * cfin[:, :, :-1])            # Scharfetter-Gummel flux
            jdif = -Dh * np.diff(cfin, axis=2)
            i_el = unit_F * (z[None, :, None] * jtot).sum(axis=1)                      # A/m^2
            zero = np.zeros((B, 1))
            dphi_iR = np.concatenate([zero, np.cumsum(np.where(kappa > 0, -i_el / kappa, 0.0) * h[None, :], axis=1)], axis=1)
            dphi_diff = np.concatenate([zero, np.cumsum(np.where(kappa > 0, unit_F * (z[None, :, None] * jdif).sum(axis=1) / kappa, 0.0)
                                                         * h[None, :], axis=1)], axis=1)
        out.update(gamma=gamma, jwall=jwall, pH=pH, kappa=kappa, i_el=i_el, dphi_iR=dphi_iR, dphi_diff=dphi_diff)
        return out

    def _alldata_fill(self, first, A, status):
        """tp.alldata[first ...] from the arrays of _alldata_arrays (the entries are rows of those arrays)."""
        tp = self.tp
        cfin, v, efield, rho = A['cfin'], A['v'], A['efield'], A['rho']
        B = cfin.shape[0]
        keys = list(tp.descriptors.keys())
        names = list(tp.species.keys())
        for b in range(B):
            i = first + b
            lane = tp.alldata_names[i]
            d = tp.alldata[i]
            for k, sp in enumerate(names):
                d['species'][sp] = {'concentration': cfin[b, k], 'surface_concentration': float(cfin[b, k, 0])}
            d['system'] = {'potential': v[b], 'efield': efield[b], 'charge_density': rho[b],
                           'surface_potential': float(v[b, 0]), 'surface_efield': float(efield[b, 0]),
                           keys[0]: lane[0], keys[1]: lane[1], 'status': int(status[b])}
        if not self.physical:
            return
        gamma, jwall, pH, kappa, i_el, dphi_iR, dphi_diff = (A[k] for k in ('gamma', 'jwall', 'pH', 'kappa', 'i_el', 'dphi_iR', 'dphi_diff'))
        ers = getattr(tp, 'electrode_reactions', None) or {}
        current = {}
        for k, sp in enumerate(names):
            if sp in ers and 'nel' in ers[sp]:          # mA/cm^2, comsol_reader.py:241-246
                nprod = len([a for a in ers[sp]['reaction'][1] if a == sp])
                current[sp] = (k, ers[sp]['nel'], nprod)
        es = tp.system.get('Stern epsilon', None)
        for b in range(B):
            d = tp.alldata[first + b]
            for k, sp in enumerate(names):
                ds = d['species'][sp]
                ds['activity_coefficient'] = gamma[b].copy()
                ds['surface_activity_coefficient'] = float(gamma[b, 0])
                ds['electrode_flux'] = float(jwall[b, k])
            for sp, (k, nel, nprod) in current.items():
                d['species'][sp]['electrode_current_density'] = float(jwall[b, k]) * nel * unit_F / nprod / 10.
            dsys = d['system']
            if pH is not None:
                dsys['pH'] = pH[b]
                dsys['surface_pH'] = float(pH[b, 0])
            dsys['activity_coefficient'] = gamma[b]
            dsys.update({'conductivity': kappa[b], 'electrolyte_current_density': i_el[b], 'delta_phi_iR': dphi_iR[b],
                         'delta_phi_diff': dphi_diff[b], 'delta_phi_iR_inf': float(dphi_iR[b, -1]),
                         'delta_phi_diff_inf': float(dphi_diff[b, -1]), 'delta_phi_inf': float(v[b, -1] - v[b, 0]),
                         'delta_phi_inf_min_iR': float(v[b, -1] - v[b, 0] - dphi_iR[b, -1])})
            if isinstance(es, (int, float)) and es:
                dsys['Stern_efield'] = float(efield[b, 0]) * tp.system['epsilon'] / es
                dsys['Stern_epsilon_func'] = es
            elif es == 'Booth':           # field-dependent Stern permittivity, comsol_reader.py:102-119, :262-273
                from .host import booth_stern_field
                dsys['Stern_efield'], dsys['Stern_epsilon_func'] = booth_stern_field(efield[b, 0], tp.system['epsilon'])

    # ------------------------------------------------------------------------------------------
    def _lane_inputs(self):
        """Per-lane pb / vzeta for the descriptor lattice (calculator.py:204-219)."""
        tp = self.tp
        keys = list(tp.descriptors.keys())
        lanes = tp.alldata_names
        B = len(lanes)
        pb = np.zeros((B, 4)); vz = np.zeros(B); phiM = np.zeros(B)
        for i, (v1, v2) in enumerate(lanes):
            system = dict(tp.system)
            system[keys[0]] = v1
            system[keys[1]] = v2
            if 'phiM' in keys and getattr(tp, 'vzeta_follows_phiM', True):
                system['vzeta'] = system['phiM']
            pb[i] = tp.pb_array(system)
            vz[i] = system['vzeta']
            phiM[i] = system['phiM']
        return pb, vz, phiM

    def initialize_surface_concentrations_from_file(self, fname, desc=None):
        """calculator.py:242-258: start the SCF loop from the surface concentrations of an earlier results folder (nine pickles,
        results_io.save_all; same descriptor list).  desc = a phiM value: that descriptor point's surface state becomes
        tp.species[sp]['surface_concentration'] -- the reference's behaviour, every lane starts there.  desc = None (batched
        extension): every lane starts from ITS OWN descriptor point of the folder."""
        from .results_io import read_all
        tp = self.tp
        holder = type('_Holder', (), {})()
        read_all(holder, fname, only=['alldata'])
        names = list(tp.species.keys())
        if desc is not None:
            inx = list(tp.descriptors['phiM']).index(desc)
            for sp in names:
                tp.species[sp]['surface_concentration'] = float(holder.alldata[inx]['species'][sp]['surface_concentration'])
            self.surface_init = None
        else:
            if len(holder.alldata) != len(tp.alldata_names):
                raise CalculatorError('initialize_surface_concentrations_from_file: %d descriptor points in %s, %d here'
                                      % (len(holder.alldata), fname, len(tp.alldata_names)))
            self.surface_init = np.array([[float(holder.alldata[i]['species'][sp]['surface_concentration']) for sp in names]
                                          for i in range(len(tp.alldata_names))])

    def run_scf_cycle(self, flux_callback=None, nel=None, nprod=None, max_iter=1000, transport_fn=None, label=''):
        """Batched counterpart of the reference's SCF outer loop (catint/calculator.py:294-406): kinetics
        (`flux_callback`, the seam where CatMAP sat, catmap_wrapper.py:106) <-> transport, once per iteration, for
        every descriptor point at the same time.  Each lane carries its own mixing factor, iteration bookkeeping
        and convergence flag; converged lanes are frozen (their surface state and fluxes stop changing) but keep
        riding in the batch.

        flux_callback(state) -> flux [B][N] in mol m^-2 s^-1 (educts negative, calculator.py:415-432), where
        state = {'surface_concentration' [B][N], 'surface_pH' [B], 'phiM' [B], 'surface_potential' [B],
                 'surface_efield' [B], 'istep'}.
        transport_fn(flux) -> (csurf [B][N], vsurf [B], esurf [B]) replaces the GPU solve (host-logic tests).
        flux_callback=None (physical mode, stationary): the kinetic model is the analytic table of set_surface_kinetics,
        evaluated EXPLICITLY from the mixed surface concentrations, and the loop runs on the device from its second
        iteration on (pnp_scf_cycle: no host round trip per iteration, converged lanes leave the batch).
        Same iterates as this host loop with surface_kinetic_fluxes(clip=True) as the callback; the hand-over happens after the
        first iteration (>= 2) in which every transport solve converged, and a lane whose solve fails later goes back to the
        state of its last converged solve instead of triggering a whole-batch restart from the bulk state.
        label: the reference's run_scf_cycle(label='') argument (calculator.py:294; it only names COMSOL's files) -- accepted, also
        as the first positional argument, and kept in self.scf_label.
        Returns a dict with the per-lane results and bookkeeping."""
        if isinstance(flux_callback, str):          # reference call shape: run_scf_cycle(label)
            flux_callback, label = None, flux_callback
        self.scf_label = label
        tp = self.tp
        device_loop = flux_callback is None
        kinetics = getattr(self, 'surface_kinetics', None)
        if device_loop:
            if not (self.physical and self.mode == 'stationary' and kinetics and transport_fn is None):
                raise CalculatorError('run_scf_cycle without a callback needs the physical mode (stationary) and set_surface_kinetics')
            flux_callback = lambda state: self.surface_kinetic_fluxes(state['surface_concentration'], state['phiM'], clip=True,  # noqa: E731
                                                                      reactions=kinetics, vsurf=state['surface_potential'])
        names = list(tp.species.keys())
        N = tp.nspecies
        pb, vz, phiM = self._lane_inputs()
        B = len(phiM)
        nel = np.ones(N) if nel is None else np.asarray(nel, float)
        nprod = np.ones(N) if nprod is None else np.asarray(nprod, float)
        if tp.system.get('init_folder') is not None:       # calculator.py:303-309: surface state of a COMSOL(-layout) results folder
            from .results_io import read_surface_concentrations
            for sp, v in zip(names, read_surface_concentrations(tp, tp.system['init_folder'])):
                tp.species[sp]['surface_concentration'] = float(v)
        sc = np.repeat(np.array([[tp.species[sp]['surface_concentration'] for sp in names]], float), B, axis=0)
        if getattr(self, 'surface_init', None) is not None:      # per-lane start (initialize_surface_concentrations_from_file)
            init = np.array(self.surface_init, float)
            self.surface_init = None          # consumed once, as the reference re-initialises once (calculator.py:242-258)
            if init.size != B * N:
                raise CalculatorError('surface concentrations read from a results folder hold %d values, the descriptor lattice of this '
                                      'calculator needs %d lanes x %d species' % (init.size, B, N))
            sc = init.reshape(B, N)
        flux = np.repeat(tp.flux_bound[None, :, 0], B, axis=0).astype(float)
        mix = np.full(B, float(self.mix_scf))
        acc = np.full(B, np.inf)
        step_to_check = np.zeros(B, int)
        active = np.ones(B, bool)
        failed = np.zeros(B, bool)      # NaN in the transport solve: the reference's nan_in_surface() (calculator.py:409-414)
        surface_pH = np.full(B, float(tp.system.get('bulk_pH', 7.0)))
        vsurf = phiM.copy(); esurf = np.zeros(B)
        sc_old = sc.copy(); cd_old = None
        iH = names.index('H+') if 'H+' in names else None
        iOH = names.index('OH-') if 'OH-' in names else None
        solver = None
        c0 = np.repeat(tp.c0[None, :], B, axis=0)
        if transport_fn is None:
            solver = self._physical_solver(B) if self.physical else self._solver(B, pb_mode_from_bound(pb[0]), tp.dx, tp.nx, tp.dt)
        istep = 0
        history = []
        restart = False
        try:
            if device_loop:
                self.surface_kinetics = None      # explicit coupling: the transport solves see prescribed fluxes only
            while active.any() and istep < max_iter:       # :316
                istep += 1
                dec = active & (istep - step_to_check > 40)                                  # :319-323
                mix[dec] *= 0.9
                step_to_check[dec] = istep
                if istep > 2:                                                               # :328-338
                    mixed = np.where(sc < 0.0, sc_old, mix[:, None] * sc + (1. - mix[:, None]) * sc_old)
                else:                                                                       # :341-344
                    mixed = np.where(sc < 0.0, 1e-20, sc)
                sc = np.where(active[:, None], mixed, sc)
                if iH is not None:                                                          # :346-359
                    ok = active & (sc[:, iH] > 0.0)
                    surface_pH[ok] = -np.log10(sc[ok, iH] / 1000.)
                elif iOH is not None:
                    ok = active & (sc[:, iOH] > 0.0)
                    surface_pH[ok] = 14 + np.log10(sc[ok, iOH] / 1000.)
                sc_old = np.where(active[:, None], sc, sc_old)                              # :362-364
                newflux = np.asarray(flux_callback({'surface_concentration': sc.copy(), 'surface_pH': surface_pH.copy(),
                                                    'phiM': phiM, 'surface_potential': vsurf.copy(),
                                                    'surface_efield': esurf.copy(), 'istep': istep}), float)
                flux = np.where(active[:, None], newflux.reshape(B, N), flux)               # :373
                if transport_fn is not None:                                                # :385 run_single_step
                    cs, vs, es = transport_fn(flux)
                    status = np.zeros(B, np.int32)
                elif self.physical:
                    # warm start from the previous SCF iterate (restart=True of comsol.run, calculator.py:523)
                    warm = istep > 1 and self.mode == 'stationary' and not restart
                    if warm:      # fluxes in, surface state out, one device synchronisation (pnp_solve_surface)
                        cs, vs, es, status = solver.solve_surface(self.RF * flux)
                    else:
                        status = self.solve_physical(solver, c0, phiM, flux, nramp=1 if istep > 1 else 8, warm=False, retry=False)
                        cs, vs, es = solver.get_surface()
                    # Wall fluxes that would drive a concentration negative have no solution inside the positive cone the
                    # damped Newton stays in: COMSOL hands the SCF loop a negative surface concentration there
                    # (calculator.py:328-344 falls back to the previous iterate); a lane that did not converge is reported
                    # the same way, and the next solve starts again from the bulk state
                    stuck = status == 1
                    cs = np.where(stuck[:, None], -1.0, cs)
                    restart = bool(stuck.any() or (status == 2).any())
                    status = np.where(stuck, 0, status)
                else:
                    solver.set_batch(c0, pb, vz, self.RF * flux)
                    n_first = 1 if self.calc == 'Crank-Nicolson' else 0
                    solver.step(tp.nt - n_first)
                    cs, vs, es = solver.get_surface()
                    status = solver.get_status()
                sc = np.where(active[:, None], cs, sc)
                vsurf = np.where(active, vs, vsurf)
                esurf = np.where(active, es, esurf)
                cd = flux * nel[None, :] * unit_F / nprod[None, :] / 10.                    # mA/cm^2, :389-400
                if istep > 1:                                                               # :401-402, signed quirk :274
                    with np.errstate(divide='ignore', invalid='ignore'):
                        err = np.where(cd != 0, np.abs(cd - cd_old) / cd, -np.inf)
                    acc = np.where(active, err.max(axis=1), acc)
                cd_old = cd if cd_old is None else np.where(active[:, None], cd, cd_old)
                history.append(acc.copy())
                bad = active & (~np.isfinite(cs).all(axis=1) | (status == 2))
                failed |= bad
                active = active & ~bad & ((acc > self.tau_scf) | (sc < 0.0).any(axis=1))
                if device_loop and active.any() and istep < max_iter and istep >= 2 and not restart and self.RF == 1.0:
                    # (with a roughness factor the loop stays on the host: the device table would fold RF into the fluxes the
                    #  bookkeeping of the loop sees, the reference scales only what COMSOL is given)
                    # hand the loop state over; the table with the rate constants at the full potentials is the kinetic model
                    self.surface_kinetics = kinetics
                    self._apply_surface_kinetics(solver, phiM)
                    st = {'surface_concentration': sc, 'surface_concentration_old': sc_old, 'flux': flux,
                          'current_density_old': cd_old, 'mix': mix, 'accuracy': acc, 'surface_pH': surface_pH,
                          'surface_potential': vsurf, 'surface_efield': esurf, 'step_to_check': step_to_check,
                          'active': active.astype(np.int32), 'failed': failed.astype(np.int32)}
                    istep = solver.scf_cycle(st, istep, max_iter, self.tau_scf, unit_F, nel, nprod,
                                             species_H=-1 if iH is None else iH, species_OH=-1 if iOH is None else iOH)
                    sc, sc_old, flux, cd_old = (st[k] for k in ('surface_concentration', 'surface_concentration_old', 'flux',
                                                                'current_density_old'))
                    mix, acc, surface_pH, vsurf, esurf = (st[k] for k in ('mix', 'accuracy', 'surface_pH', 'surface_potential',
                                                                          'surface_efield'))
                    step_to_check = st['step_to_check']
                    active = st['active'] != 0
                    failed = st['failed'] != 0
                    break
        finally:
            if device_loop:
                self.surface_kinetics = kinetics
            if solver is not None:
                solver.close()
        for i in range(B):
            d = tp.alldata[i]
            d.setdefault('species', {}); d.setdefault('system', {})
            for k, sp in enumerate(names):
                d['species'].setdefault(sp, {})
                d['species'][sp]['surface_concentration'] = float(sc[i, k])
                d['species'][sp]['flux'] = float(flux[i, k])
                d['species'][sp]['electrode_current_density'] = float(flux[i, k] * nel[k] * unit_F / nprod[k] / 10.)
            d['system'].update({'surface_pH': float(surface_pH[i]), 'surface_potential': float(vsurf[i]),
                                'surface_efield': float(esurf[i]), 'phiM': float(phiM[i])})
        return {'surface_concentration': sc, 'flux': flux, 'current_density': flux * nel * unit_F / nprod / 10.,
                'surface_pH': surface_pH, 'accuracy': acc, 'converged': ~active & ~failed, 'failed': failed,
                'iterations': istep, 'mix': mix,
                'history': np.array(history)}

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def evaluate_accuracy(par, par_old):
        """calculator.py:260-283 -- note the division by the SIGNED new value."""
        acc = -np.inf
        for k in par:
            p1, p2 = par[k], par_old[k]
            if p1 != 0:
                acc = max(acc, abs(p1 - p2) / p1)
        return acc
