"""ctypes binding of the C-ABI in ``include/catint_pnp.h`` (the drop-in boundary).

The shared library is built in-tree by ``__graft_entry__.build()`` /
``catint_amd.build.build_library()`` into ``catint_amd/lib/libcatint_pnp.so``.  There is no
CPU fallback: if the library is missing, or no HIP device is visible, the transport path
raises (``PnpLibraryError`` / ``PnpError``).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CATINT_PNP_LIB') or os.path.join(_HERE, 'lib', 'libcatint_pnp.so')   # (env: diagnosis builds of tools/probe)

PNP_MAX_SPECIES = 16
PNP_MAX_REACTIONS = 16
PNP_MAX_REACTANTS = 4

METHOD_CRANK_NICOLSON = 0
METHOD_FTCS = 1
METHOD_NEWTON = 2      # physical mode: fully implicit coupled Newton (what the reference asks COMSOL for)
METHODS = {'Crank-Nicolson': METHOD_CRANK_NICOLSON, 'FTCS': METHOD_FTCS, 'Newton': METHOD_NEWTON}
PNP_NEWTON_MAX_SPECIES = 8

PB_DD, PB_VWALL_GBULK, PB_GWALL_VBULK, PB_VWALL_GWALL, PB_VBULK_GBULK = range(5)

STATUS_OK, STATUS_MAXIT, STATUS_NAN, STATUS_NEGATIVE = 0, 1, 2, 3

# every symbol include/catint_pnp.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    'pnp_create', 'pnp_destroy', 'pnp_last_error', 'pnp_version', 'pnp_set_species', 'pnp_set_reactions',
    'pnp_set_batch', 'pnp_set_flux', 'pnp_set_pb', 'pnp_step', 'pnp_integrate', 'pnp_mol_rhs', 'pnp_integrate_dopri5', 'pnp_integrate_dop853', 'pnp_integrate_rkc', 'pnp_get_state',
    'pnp_get_surface', 'pnp_get_status', 'pnp_synchronize', 'pnp_timer_start', 'pnp_timer_stop',
    'pnp_device_bytes', 'pnp_row_pitch', 'pnp_step_row_chunks', 'pnp_set_newton', 'pnp_solve_stationary', 'pnp_get_newton_iterations',
    'pnp_set_potential', 'pnp_set_convection', 'pnp_set_option', 'pnp_get_lane_order', 'pnp_autotune', 'pnp_autotune_name', 'pnp_autotune_default', 'pnp_tune_placement', 'pnp_set_lanes', 'pnp_set_lane_mask', 'pnp_set_wall_kinetics', 'pnp_set_wall_rate_law', 'pnp_set_grid', 'pnp_solve_surface', 'pnp_scf_cycle',
]


class PnpLibraryError(RuntimeError):
    pass


class PnpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('catint_pnp error %d: %s' % (code, msg))
        self.code = code


class PnpConfig(C.Structure):
    _fields_ = [
        ('struct_size', C.c_int32), ('device', C.c_int32), ('nspecies', C.c_int32), ('nx', C.c_int32),
        ('method', C.c_int32), ('pb_mode', C.c_int32), ('lax_friedrich', C.c_int32), ('use_migration', C.c_int32),
        ('batch_capacity', C.c_int64), ('dx', C.c_double), ('dt', C.c_double), ('beta', C.c_double),
        ('eps', C.c_double),
    ]


class PnpNewtonParams(C.Structure):
    _fields_ = [
        ('struct_size', C.c_int32), ('wall_bc', C.c_int32), ('maxit', C.c_int32), ('error_estimate', C.c_int32),
        ('stern_capacitance', C.c_double), ('phi_pzc', C.c_double), ('tol', C.c_double), ('dphi_max', C.c_double),
        ('time_order', C.c_int32), ('predictor', C.c_int32),
    ]


class PnpScfParams(C.Structure):
    _fields_ = [
        ('struct_size', C.c_int32), ('istep', C.c_int32), ('max_iter', C.c_int32), ('check_every', C.c_int32),
        ('species_H', C.c_int32), ('species_OH', C.c_int32), ('tau_scf', C.c_double), ('faraday', C.c_double),
    ]


class PnpOdeParams(C.Structure):
    _fields_ = [
        ('struct_size', C.c_int32), ('nsteps', C.c_int32), ('rtol', C.c_double), ('atol', C.c_double),
        ('first_step', C.c_double), ('max_step', C.c_double), ('safety', C.c_double), ('ifactor', C.c_double),
        ('dfactor', C.c_double), ('beta', C.c_double), ('nstiff', C.c_int32), ('check_every', C.c_int32),
    ]


class PnpScfState(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_double)) for n in (
        'surface_concentration', 'surface_concentration_old', 'flux', 'current_density_old', 'mix', 'accuracy',
        'surface_pH', 'surface_potential', 'surface_efield')] + [(n, C.POINTER(C.c_int32)) for n in (
            'step_to_check', 'active', 'failed')]


_lib = None


def load_library():
    """dlopen the HIP library; raises PnpLibraryError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PnpLibraryError(
            'HIP extension %s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950). There is no CPU fallback for the transport path.' % LIB_PATH)
    if os.path.exists(os.path.join(_HERE, 'lib', '.partial')) and not os.environ.get('CATINT_ALLOW_PARTIAL'):   # dev runs set it
        raise PnpLibraryError(
            '%s was built by tools/devbuild.sh with only one Newton block size instantiated (marker lib/.partial): rebuild the '
            'full library with `python -c "import __graft_entry__ as g; g.build()"`.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    vp = C.c_void_p
    lib.pnp_create.argtypes = [C.POINTER(PnpConfig), C.POINTER(vp)]
    lib.pnp_create.restype = C.c_int
    lib.pnp_destroy.argtypes = [vp]
    lib.pnp_destroy.restype = None
    lib.pnp_last_error.argtypes = [vp]
    lib.pnp_last_error.restype = C.c_char_p
    lib.pnp_version.argtypes = []
    lib.pnp_version.restype = C.c_char_p
    lib.pnp_set_species.argtypes = [vp, dp, dp]
    lib.pnp_set_reactions.argtypes = [vp, C.c_int32, ip, ip, ip, ip, dp, dp]
    lib.pnp_set_batch.argtypes = [vp, C.c_int64, dp, dp, dp, dp]
    lib.pnp_set_flux.argtypes = [vp, dp]
    lib.pnp_set_pb.argtypes = [vp, dp, dp]
    lib.pnp_step.argtypes = [vp, C.c_int32, C.c_int32]
    lib.pnp_integrate.argtypes = [vp, C.c_int32, ip, C.c_int32, dp, ip]
    lib.pnp_mol_rhs.argtypes = [vp, dp, dp]
    lib.pnp_get_state.argtypes = [vp, dp, dp, dp, dp]
    lib.pnp_get_surface.argtypes = [vp, dp, dp, dp]
    lib.pnp_get_status.argtypes = [vp, ip]
    lib.pnp_synchronize.argtypes = [vp]
    lib.pnp_timer_start.argtypes = [vp]
    lib.pnp_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    lib.pnp_set_newton.argtypes = [vp, C.POINTER(PnpNewtonParams), dp]
    lib.pnp_solve_stationary.argtypes = [vp, C.c_double, C.c_int32, ip]
    lib.pnp_get_newton_iterations.argtypes = [vp, ip]
    lib.pnp_set_potential.argtypes = [vp, dp]
    lib.pnp_set_lanes.argtypes = [vp, C.c_int64, C.POINTER(C.c_int64), dp, dp]
    lib.pnp_set_lanes.restype = C.c_int
    lib.pnp_set_lane_mask.argtypes = [vp, ip]
    lib.pnp_set_lane_mask.restype = C.c_int
    lib.pnp_set_wall_kinetics.argtypes = [vp, C.c_int32, ip, dp, dp]
    lib.pnp_set_wall_kinetics.restype = C.c_int
    lib.pnp_integrate_dopri5.argtypes = [vp, C.POINTER(PnpOdeParams), C.c_int32, ip, C.c_int32, dp, ip, C.POINTER(C.c_int64), dp]
    lib.pnp_integrate_dopri5.restype = C.c_int
    lib.pnp_integrate_dop853.argtypes = lib.pnp_integrate_dopri5.argtypes
    lib.pnp_integrate_dop853.restype = C.c_int
    lib.pnp_integrate_rkc.argtypes = lib.pnp_integrate_dopri5.argtypes
    lib.pnp_integrate_rkc.restype = C.c_int
    lib.pnp_set_wall_rate_law.argtypes = [vp, C.c_int32, dp, dp]
    lib.pnp_set_wall_rate_law.restype = C.c_int
    lib.pnp_set_grid.argtypes = [vp, dp]
    lib.pnp_set_grid.restype = C.c_int
    lib.pnp_set_convection.argtypes = [vp, C.c_double]
    lib.pnp_set_convection.restype = C.c_int
    lib.pnp_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.pnp_set_option.restype = C.c_int
    lib.pnp_get_lane_order.argtypes = [vp, ip]
    lib.pnp_get_lane_order.restype = C.c_int64
    lib.pnp_autotune.argtypes = [vp, C.c_int32, dp, ip]
    lib.pnp_autotune.restype = C.c_int
    lib.pnp_autotune_name.argtypes = [C.c_int32]
    lib.pnp_autotune_name.restype = C.c_char_p
    lib.pnp_autotune_default.argtypes = [vp]
    lib.pnp_autotune_default.restype = C.c_int32
    lib.pnp_tune_placement.argtypes = [vp, C.c_int32, C.c_int32, dp]
    lib.pnp_tune_placement.restype = C.c_int
    lib.pnp_solve_surface.argtypes = [vp, dp, C.c_int32, dp, dp, dp, ip]
    lib.pnp_solve_surface.restype = C.c_int
    lib.pnp_scf_cycle.argtypes = [vp, C.POINTER(PnpScfParams), dp, dp, C.POINTER(PnpScfState), ip]
    lib.pnp_scf_cycle.restype = C.c_int
    for name in ('pnp_set_newton', 'pnp_solve_stationary', 'pnp_get_newton_iterations', 'pnp_set_potential'):
        getattr(lib, name).restype = C.c_int
    for name in ('pnp_set_species', 'pnp_set_reactions', 'pnp_set_batch', 'pnp_set_flux', 'pnp_set_pb', 'pnp_step',
                 'pnp_integrate', 'pnp_mol_rhs', 'pnp_get_state', 'pnp_get_surface', 'pnp_get_status', 'pnp_synchronize',
                 'pnp_timer_start', 'pnp_timer_stop'):
        getattr(lib, name).restype = C.c_int
    lib.pnp_device_bytes.argtypes = [vp]
    lib.pnp_device_bytes.restype = C.c_int64
    lib.pnp_row_pitch.argtypes = [vp]
    lib.pnp_row_pitch.restype = C.c_int32
    lib.pnp_step_row_chunks.argtypes = [vp, C.c_int32]
    lib.pnp_step_row_chunks.restype = C.c_int32
    _lib = lib
    return lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError('expected shape %s, got %s' % (tuple(shape), a.shape))
    return a


class PnpSolver(object):
    """Thin object wrapper over one ``pnp_handle`` (one GPU)."""

    def __init__(self, nspecies, nx, dx, dt, beta, eps, D, charges, method='Crank-Nicolson', pb_mode=PB_DD,
                 lax_friedrich=False, use_migration=True, batch_capacity=1, device=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        if method not in METHODS:
            raise PnpError(-1, 'No calculator found with this name: %r' % (method,))
        cfg = PnpConfig(C.sizeof(PnpConfig), int(device), int(nspecies), int(nx), METHODS[method], int(pb_mode),
                        int(bool(lax_friedrich)), int(bool(use_migration)), int(batch_capacity), float(dx), float(dt),
                        float(beta), float(eps))
        rc = self._lib.pnp_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = self._lib.pnp_last_error(None).decode()
            self._h = C.c_void_p()
            raise PnpError(rc, msg)
        self.N, self.nx, self.B = int(nspecies), int(nx), 0
        self.dt_ode = float(dt)      # interval length of integrate_dopri5 (cfg.dt)
        self.method = method
        self._check(self._lib.pnp_set_species(self._h, _dptr(_f64(D, (self.N,))), _dptr(_f64(charges, (self.N,)))))

    def _check(self, rc):
        if rc != 0:
            raise PnpError(rc, self._lib.pnp_last_error(self._h).decode())

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.pnp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- parameters ------------------------------------------------------------------------
    def set_reactions(self, reactions):
        """reactions: list of (lhs_indices, rhs_indices, kf, kr) in the reference's dict order."""
        n = len(reactions)
        n_lhs = np.zeros(max(n, 1), np.int32)
        n_rhs = np.zeros(max(n, 1), np.int32)
        lhs = np.zeros((max(n, 1), PNP_MAX_REACTANTS), np.int32)
        rhs = np.zeros((max(n, 1), PNP_MAX_REACTANTS), np.int32)
        kf = np.zeros(max(n, 1))
        kr = np.zeros(max(n, 1))
        for r, (l, rr, f, b) in enumerate(reactions):
            if len(l) > PNP_MAX_REACTANTS or len(rr) > PNP_MAX_REACTANTS:
                raise PnpError(-1, 'too many reactants in reaction %d' % r)
            n_lhs[r], n_rhs[r] = len(l), len(rr)
            lhs[r, :len(l)] = l
            rhs[r, :len(rr)] = rr
            kf[r], kr[r] = f, b
        self._check(self._lib.pnp_set_reactions(self._h, n, _iptr(n_lhs), _iptr(lhs), _iptr(n_rhs), _iptr(rhs),
                                                _dptr(kf), _dptr(kr)))

    def set_batch(self, c0, pb, vzeta, flux):
        c0 = np.ascontiguousarray(c0, dtype=np.float64)
        B = c0.shape[0]
        c0 = _f64(c0.reshape(B, self.N, self.nx))
        pb = np.nan_to_num(_f64(pb, (B, 4)), nan=0.0)
        self._check(self._lib.pnp_set_batch(self._h, B, _dptr(c0), _dptr(pb), _dptr(_f64(vzeta, (B,))),
                                            _dptr(_f64(flux, (B, self.N)))))
        self.B = B

    def set_flux(self, flux):
        self._check(self._lib.pnp_set_flux(self._h, _dptr(_f64(flux, (self.B, self.N)))))

    def set_pb(self, pb, vzeta):
        pb = np.nan_to_num(_f64(pb, (self.B, 4)), nan=0.0)
        self._check(self._lib.pnp_set_pb(self._h, _dptr(pb), _dptr(_f64(vzeta, (self.B,)))))

    # -- physical mode ---------------------------------------------------------------------
    def set_newton(self, wall_bc='dirichlet', stern_capacitance=0.0, phi_pzc=0.0, tol=1e-10, maxit=50, dphi_max=0.05,
                   mpb_radius=None, error_estimate=False, time_order=1, predictor=False):
        """Boundary model and Newton controls of the physical mode (tp.system['Stern capacitance'], ['phiPZC'],
        tp.species[sp]['MPB_radius']; comsol_model.py:613,:982,:1041-1063).  time_order=2: BDF2 timesteps (the reference's transient
        study: BDF, maxorder 2, comsol_model.py:518-531) instead of backward Euler; predictor=True: every timestep's Newton iteration
        starts from the linear extrapolation of the two previous time levels."""
        p = PnpNewtonParams(C.sizeof(PnpNewtonParams), {'dirichlet': 0, 'stern': 1}[wall_bc], int(maxit), int(bool(error_estimate)),
                            float(stern_capacitance), float(phi_pzc), float(tol), float(dphi_max), int(time_order), int(bool(predictor)))
        r = None if mpb_radius is None else _f64(mpb_radius, (self.N,))
        self._check(self._lib.pnp_set_newton(self._h, C.byref(p), _dptr(r)))

    def set_option(self, key, value):
        """Debug / tuning switch of this handle (pnp_set_option): e.g. set_option('NEWTON_KERNEL', 'lane2').  The CATINT_* environment
        variables only provide the defaults at construction."""
        self._check(self._lib.pnp_set_option(self._h, str(key).encode(), str(value).encode()))

    def lane_order(self):
        """Debug: operating point of every slot of the most recent lane-kernel launch (pnp_get_lane_order); empty if none was used."""
        n = int(self._lib.pnp_get_lane_order(self._h, None))
        perm = np.zeros(max(n, 0), np.int32)
        if n > 0:
            self._lib.pnp_get_lane_order(self._h, _iptr(perm))
        return perm

    def default_family(self):
        """Name of the kernel family the library's thresholds choose for the current batch (pnp_autotune_default); None without one."""
        i = int(self._lib.pnp_autotune_default(self._h))
        return None if i < 0 else self._lib.pnp_autotune_name(i).decode()

    def tune_placement(self, nsteps=2, trials=4):
        """Lane kernels: allocate the workspace up to `trials` times, keep the placement on which `nsteps` timesteps run fastest
        (pnp_tune_placement); the state is left as it was.  Returns the time per timestep (ms) of the trials that were made."""
        ms = np.full(int(trials), -1.0)
        self._check(self._lib.pnp_tune_placement(self._h, int(nsteps), int(trials), _dptr(ms)))
        return [float(v) for v in ms if v >= 0.0]

    def autotune(self, nsteps=2):
        """Physical mode: pick the kernel family by measurement on this device and batch (pnp_autotune).  Returns (name of the chosen
        family, {family: ms per timestep} of every family that supports the shape); the state and its history are left as they were."""
        n = 8
        ms = np.zeros(n)
        chosen = np.zeros(1, np.int32)
        self._check(self._lib.pnp_autotune(self._h, int(nsteps), _dptr(ms), _iptr(chosen)))
        names = [self._lib.pnp_autotune_name(i).decode() for i in range(n)]
        return names[int(chosen[0])], {names[i]: float(ms[i]) for i in range(n) if ms[i] >= 0.0}

    def set_convection(self, velocity):
        """Constant convection velocity (m/s) of the physical mode: tp.system['flow rate'] (comsol_model.py:901-903)."""
        self._check(self._lib.pnp_set_convection(self._h, float(velocity)))

    def set_grid(self, x):
        """Non-uniform grid x[nx] of the physical mode (electrode at x[0]); dx of the constructor stays the scaling length."""
        self._check(self._lib.pnp_set_grid(self._h, _dptr(_f64(x, (self.nx,)))))

    def set_wall_kinetics(self, species, nu, k, alpha=None, saturation=None):
        """Surface reactions coupled implicitly: species [n] (index, -1 = zeroth order), nu [n][N] stoichiometry of the flux
        into the domain, k [B][n] rate constants per lane.  Empty lists remove the table.  alpha [n] (1/V) / saturation [n]
        (m^3/mol): Butler-Volmer factor exp(alpha (phiM - phi(x=0))) and Langmuir saturation c/(1 + K c) (pnp_set_wall_rate_law);
        None: first order."""
        n = len(species)
        if n == 0:
            self._check(self._lib.pnp_set_wall_kinetics(self._h, 0, None, None, None))
            return
        sp = np.ascontiguousarray(species, dtype=np.int32)
        self._check(self._lib.pnp_set_wall_kinetics(self._h, n, _iptr(sp), _dptr(_f64(nu, (n, self.N))),
                                                    _dptr(_f64(k, (self.B, n)))))
        if alpha is not None or saturation is not None:
            al = _f64(np.zeros(n) if alpha is None else alpha, (n,))
            ks = _f64(np.zeros(n) if saturation is None else saturation, (n,))
            self._check(self._lib.pnp_set_wall_rate_law(self._h, n, _dptr(al), _dptr(ks)))

    def solve_stationary(self, tol=0.0, maxit=0):
        st = np.zeros(self.B, np.int32)
        self._check(self._lib.pnp_solve_stationary(self._h, float(tol), int(maxit), _iptr(st)))
        return st

    def solve_surface(self, flux=None, nsteps=0):
        """New wall fluxes in, surface state out, one synchronisation: (csurf [B][N], vsurf [B], esurf [B], status [B]);
        nsteps = 0 solves the stationary problem, otherwise nsteps backward-Euler steps."""
        cs = np.zeros((self.B, self.N)); vs = np.zeros(self.B); es = np.zeros(self.B); st = np.zeros(self.B, np.int32)
        f = None if flux is None else _f64(flux, (self.B, self.N))
        self._check(self._lib.pnp_solve_surface(self._h, _dptr(f), int(nsteps), _dptr(cs), _dptr(vs), _dptr(es), _iptr(st)))
        return cs, vs, es, st

    def scf_cycle(self, state, istep, max_iter, tau_scf, faraday, nel=None, nprod=None, species_H=-1, species_OH=-1,
                  check_every=8):
        """Device SCF loop (pnp_scf_cycle) from iteration istep+1 on.  `state`: dict of the loop arrays
        ('surface_concentration', 'surface_concentration_old', 'flux', 'current_density_old' [B][N]; 'mix', 'accuracy',
        'surface_pH', 'surface_potential', 'surface_efield' [B]; 'step_to_check', 'active', 'failed' [B] int) -- updated in
        place (fresh contiguous arrays are put back into the dict).  Returns the last iteration number that had active lanes."""
        B, N = self.B, self.N
        st = PnpScfState()
        for name, ctype in PnpScfState._fields_:
            if ctype == C.POINTER(C.c_double):
                shape = (B, N) if name in ('surface_concentration', 'surface_concentration_old', 'flux', 'current_density_old') else (B,)
                arr = np.array(_f64(state[name], shape))
                setattr(st, name, _dptr(arr))
            else:
                arr = np.ascontiguousarray(np.asarray(state[name]).reshape(B), dtype=np.int32).copy()
                setattr(st, name, _iptr(arr))
            state[name] = arr
        p = PnpScfParams(struct_size=C.sizeof(PnpScfParams), istep=int(istep), max_iter=int(max_iter), check_every=int(check_every),
                         species_H=int(species_H), species_OH=int(species_OH), tau_scf=float(tau_scf), faraday=float(faraday))
        nel = None if nel is None else _f64(nel, (N,))
        nprod = None if nprod is None else _f64(nprod, (N,))
        it = C.c_int32(0)
        self._check(self._lib.pnp_scf_cycle(self._h, C.byref(p), _dptr(nel), _dptr(nprod), C.byref(st), C.byref(it)))
        return int(it.value)

    def newton_iterations(self):
        it = np.zeros(self.B, np.int32)
        self._check(self._lib.pnp_get_newton_iterations(self._h, _iptr(it)))
        return it

    def set_potential(self, phi):
        self._check(self._lib.pnp_set_potential(self._h, _dptr(_f64(phi, (self.B, self.nx)))))

    def set_lanes(self, lanes, c, phi=None):
        """State of a few lanes (c [n][N][nx] or [n][N*nx], phi [n][nx] or None) without touching the rest of the batch, its counters
        or its flags (pnp_set_lanes)."""
        idx = np.ascontiguousarray(lanes, dtype=np.int64)
        n = len(idx)
        cc = _f64(np.asarray(c, float).reshape(n, self.N, self.nx), (n, self.N, self.nx))
        pp = None if phi is None else _f64(phi, (n, self.nx))
        self._check(self._lib.pnp_set_lanes(self._h, n, idx.ctypes.data_as(C.POINTER(C.c_int64)), _dptr(cc), _dptr(pp)))

    def set_lane_mask(self, mask=None):
        """Only lanes with a non-zero mask take part in the following solves of the physical mode (None: all lanes again)."""
        if mask is None:
            self._check(self._lib.pnp_set_lane_mask(self._h, None))
            return
        m = np.ascontiguousarray(np.asarray(mask).astype(np.int32).reshape(self.B))
        self._check(self._lib.pnp_set_lane_mask(self._h, _iptr(m)))

    # -- hot path --------------------------------------------------------------------------
    def step(self, nsteps=1, steps_per_launch=0):
        self._check(self._lib.pnp_step(self._h, int(nsteps), int(steps_per_launch)))

    def integrate(self, nt, itout):
        """Reference loop + outputs: returns (cout[n_out, B, N*nx], status[B])."""
        itout = np.ascontiguousarray(itout, dtype=np.int32)
        cout = np.zeros((len(itout), self.B, self.N * self.nx))
        status = np.zeros(self.B, np.int32)
        self._check(self._lib.pnp_integrate(self._h, int(nt), _iptr(itout), len(itout), _dptr(cout), _iptr(status)))
        return cout, status

    def mol_rhs(self, c):
        """ode_func for the current batch: c [B][N*nx] -> dc/dt [B][N*nx] (calculator_old.py:827-935)."""
        c = _f64(np.asarray(c, dtype=np.float64).reshape(self.B, self.N * self.nx))
        out = np.zeros_like(c)
        self._check(self._lib.pnp_mol_rhs(self._h, _dptr(c), _dptr(out)))
        return out

    def integrate_dopri5(self, nt, itout, rtol=1e-6, atol=1e-12, nsteps=500, first_step=0.0, max_step=0.0, safety=0.9, ifactor=10.0,
                         dfactor=0.2, beta=0.0, nstiff=1000, check_every=0):
        """scipy.integrate.ode('dopri5') of every lane on the device, one integrate(t + dt) per interval (calculator_old.py:955-963;
        keyword names and defaults are scipy's).  Returns (cout[n_out, B, N*nx] = state after interval itout[j], idid[B], stats[B, 5] =
        attempted / accepted / rejected steps, RHS evaluations, interval of the last call, t_end[B])."""
        return self._integrate_rk(self._lib.pnp_integrate_dopri5, nt, itout, rtol, atol, nsteps, first_step, max_step, safety, ifactor,
                                  dfactor, beta, nstiff, check_every)

    def integrate_dop853(self, nt, itout, rtol=1e-6, atol=1e-12, nsteps=500, first_step=0.0, max_step=0.0, safety=0.9, ifactor=6.0,
                         dfactor=0.3, beta=0.0, nstiff=1000, check_every=0):
        """scipy.integrate.ode('dop853') of every lane on the device (keyword names and defaults are scipy's for this integrator)."""
        return self._integrate_rk(self._lib.pnp_integrate_dop853, nt, itout, rtol, atol, nsteps, first_step, max_step, safety, ifactor,
                                  dfactor, beta, nstiff, check_every)

    def integrate_rkc(self, nt, itout, rtol=1e-6, atol=1e-12, nsteps=100000, max_step=0.0, check_every=0):
        """Stiff method-of-lines integration of every lane on the device (pnp_integrate_rkc: Runge-Kutta-Chebyshev with error control,
        the batched counterpart of odeint / ode('vode' | 'lsoda'), calculator_old.py:946-963).  Returns (cout[n_out, B, N*nx] = state
        after interval itout[j], idid[B], stats[B, 7] = attempted / accepted / rejected steps, RHS evaluations of the steps, interval
        of the last call, RHS evaluations of the spectral-radius estimates, largest stage count; t_end[B])."""
        itout = np.ascontiguousarray(itout, dtype=np.int32)
        cout = np.zeros((len(itout), self.B, self.N * self.nx))
        idid = np.zeros(self.B, np.int32)
        stats = np.zeros((self.B, 7), np.int64)
        t_end = np.zeros(self.B)
        p = PnpOdeParams(C.sizeof(PnpOdeParams), int(nsteps), float(rtol), float(atol), 0.0, float(max_step), 0.0, 0.0, 0.0, 0.0, 0,
                         int(check_every))
        self._check(self._lib.pnp_integrate_rkc(self._h, C.byref(p), int(nt), _iptr(itout), len(itout), _dptr(cout), _iptr(idid),
                                                stats.ctypes.data_as(C.POINTER(C.c_int64)), _dptr(t_end)))
        return cout, idid, stats, t_end

    def _integrate_rk(self, fn, nt, itout, rtol, atol, nsteps, first_step, max_step, safety, ifactor, dfactor, beta, nstiff, check_every):
        itout = np.ascontiguousarray(itout, dtype=np.int32)
        cout = np.zeros((len(itout), self.B, self.N * self.nx))
        idid = np.zeros(self.B, np.int32)
        stats = np.zeros((self.B, 5), np.int64)
        t_end = np.zeros(self.B)
        p = PnpOdeParams(C.sizeof(PnpOdeParams), int(nsteps), float(rtol), float(atol), float(first_step), float(max_step), float(safety),
                         float(ifactor), float(dfactor), float(beta), int(nstiff), int(check_every))
        self._check(fn(self._h, C.byref(p), int(nt), _iptr(itout), len(itout), _dptr(cout), _iptr(idid),
                       stats.ctypes.data_as(C.POINTER(C.c_int64)), _dptr(t_end)))
        return cout, idid, stats, t_end

    # -- read-back -------------------------------------------------------------------------
    def get_state(self, potential=True):
        c = np.zeros((self.B, self.N, self.nx))
        if potential:
            v = np.zeros((self.B, self.nx)); g = np.zeros((self.B, self.nx)); l = np.zeros((self.B, self.nx))
            self._check(self._lib.pnp_get_state(self._h, _dptr(c), _dptr(v), _dptr(g), _dptr(l)))
            return c, v, g, l
        self._check(self._lib.pnp_get_state(self._h, _dptr(c), None, None, None))
        return c

    def get_surface(self):
        cs = np.zeros((self.B, self.N)); vs = np.zeros(self.B); es = np.zeros(self.B)
        self._check(self._lib.pnp_get_surface(self._h, _dptr(cs), _dptr(vs), _dptr(es)))
        return cs, vs, es

    def get_status(self):
        st = np.zeros(self.B, np.int32)
        self._check(self._lib.pnp_get_status(self._h, _iptr(st)))
        return st

    def synchronize(self):
        self._check(self._lib.pnp_synchronize(self._h))

    def timer_start(self):
        self._check(self._lib.pnp_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float(0)
        self._check(self._lib.pnp_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)

    @property
    def device_bytes(self):
        return int(self._lib.pnp_device_bytes(self._h))

    @property
    def row_pitch(self):
        return int(self._lib.pnp_row_pitch(self._h))

    def step_row_chunks(self, launches):
        """Row chunks (HIP streams) a step() call of `launches` launches is cut into."""
        return int(self._lib.pnp_step_row_chunks(self._h, int(launches)))
