"""TEST INFRASTRUCTURE -- CPU restatement of the integrator behind the reference's calc='dopri5' method-of-lines path.

The reference (catint/calculator_old.py:955-963) builds scipy.integrate.ode(ode_func).set_integrator('dopri5', nsteps=10000) and
calls r.integrate(r.t + dt) once per output interval.  scipy's 'dopri5' is a wrapper around E. Hairer's DOPRI5 (Hairer, Norsett,
Wanner, Solving ODEs I, 2nd ed., II.4/II.5; dopri5.f of 1996): Dormand-Prince 5(4) with FSAL, error norm
sqrt(mean((err_i/(atol + rtol max(|y_i|, |ynew_i|)))^2)), Lund-stabilised step controller, stiffness detection.  The package is a
third-party dependency of the reference (scipy, not vendored under /root/reference; this image ships scipy 1.15.3); this file
restates its published algorithm and is pinned against scipy.integrate.ode('dopri5') itself (tests/test_ode_oracle.py: same
accepted/rejected step sequence, same trajectory) including the wrapper's observable behaviour:

  * every integrate() call is a fresh DOPRI5 call -- k1 is evaluated again, FACOLD, REJECT, the stiffness counters and the step
    counters start over -- but the step size is carried (DOPRI5 writes the predicted step back into WORK(7), which scipy keeps), so
    HINIT runs in the first call only;
  * scipy's defaults: rtol 1e-6, atol 1e-12, safety 0.9, dfactor 0.2 (FAC1), ifactor 10 (FAC2), beta 0 -> DOPRI5's default 0.04,
    max_step 0 -> XEND - X, NSTIFF 1000, UROUND 2.3e-16.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path is the device
integrator (catint_amd/csrc/pnp_ode.hip)."""
import numpy as np

C2, C3, C4, C5 = 0.2, 0.3, 0.8, 8.0 / 9.0
A21 = 0.2
A31, A32 = 3.0 / 40.0, 9.0 / 40.0
A41, A42, A43 = 44.0 / 45.0, -56.0 / 15.0, 32.0 / 9.0
A51, A52, A53, A54 = 19372.0 / 6561.0, -25360.0 / 2187.0, 64448.0 / 6561.0, -212.0 / 729.0
A61, A62, A63, A64, A65 = 9017.0 / 3168.0, -355.0 / 33.0, 46732.0 / 5247.0, 49.0 / 176.0, -5103.0 / 18656.0
A71, A73, A74, A75, A76 = 35.0 / 384.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0
E1, E3, E4, E5, E6, E7 = 71.0 / 57600.0, -71.0 / 16695.0, 71.0 / 1920.0, -17253.0 / 339200.0, 22.0 / 525.0, -1.0 / 40.0
UROUND = 2.3e-16


def _seqsum(v):
    """Left-to-right sum, as the Fortran loops accumulate (np.sum adds pairwise / in eight strands)."""
    return float(np.cumsum(v)[-1])


class Dopri5(object):
    """Stateful like scipy's ode object: set the initial value, then integrate(t_end) interval by interval."""

    def __init__(self, f, rtol=1e-6, atol=1e-12, nsteps=500, max_step=0.0, first_step=0.0, safety=0.9, ifactor=10.0, dfactor=0.2,
                 beta=0.0, nstiff=1000):
        self.f = f
        self.rtol, self.atol, self.nmax = float(rtol), float(atol), int(nsteps)
        self.max_step, self.h = float(max_step), float(first_step)
        self.safe, self.fac1, self.fac2 = float(safety), float(dfactor), float(ifactor)
        self.beta = 0.04 if beta == 0.0 else (0.0 if beta < 0.0 else float(beta))
        self.nstiff = int(nstiff)
        self.idid = 1
        self.log = []                # (x, h, err, accepted) of every attempted step
        self.nfcn = 0

    def set_initial_value(self, y, t=0.0):
        self.y, self.t = np.array(y, float), float(t)
        return self

    def successful(self):
        return self.idid >= 0

    def _hinit(self, x, y, posneg, f0, hmax):
        sk = self.atol + self.rtol * np.abs(y)
        dnf = _seqsum((f0 / sk) ** 2)
        dny = _seqsum((y / sk) ** 2)
        h = 1.0e-6 if (dnf <= 1e-10 or dny <= 1e-10) else np.sqrt(dny / dnf) * 0.01
        h = min(h, hmax) * posneg
        f1 = self.f(x + h, y + h * f0)
        der2 = np.sqrt(_seqsum(((f1 - f0) / sk) ** 2)) / h
        der12 = max(abs(der2), np.sqrt(dnf))
        h1 = max(1.0e-6, abs(h) * 1.0e-3) if der12 <= 1e-15 else (0.01 / der12) ** (1.0 / 5.0)
        return min(100 * abs(h), h1, hmax) * posneg

    def integrate(self, xend):
        f, y, x = self.f, self.y, self.t
        n = y.size
        hmax = abs(self.max_step if self.max_step != 0.0 else xend - x)
        facold, expo1 = 1.0e-4, 0.2 - self.beta * 0.75
        facc1, facc2 = 1.0 / self.fac1, 1.0 / self.fac2
        posneg = 1.0 if xend - x >= 0 else -1.0
        last, reject = False, False
        hlamb, iasti, nonsti = 0.0, 0, 0
        nstep = naccpt = 0
        k1 = f(x, y)
        h = self.h
        if h == 0.0:
            h = self._hinit(x, y, posneg, k1, hmax)
        self.nfcn += 2
        while True:
            if nstep > self.nmax:
                self.idid = -2
                break
            if 0.1 * abs(h) <= abs(x) * UROUND:
                self.idid = -3
                break
            if (x + 1.01 * h - xend) * posneg > 0.0:
                h = xend - x
                last = True
            nstep += 1
            k2 = f(x + C2 * h, y + h * A21 * k1)
            k3 = f(x + C3 * h, y + h * (A31 * k1 + A32 * k2))
            k4 = f(x + C4 * h, y + h * (A41 * k1 + A42 * k2 + A43 * k3))
            k5 = f(x + C5 * h, y + h * (A51 * k1 + A52 * k2 + A53 * k3 + A54 * k4))
            ysti = y + h * (A61 * k1 + A62 * k2 + A63 * k3 + A64 * k4 + A65 * k5)
            xph = x + h
            k6 = f(xph, ysti)
            y1 = y + h * (A71 * k1 + A73 * k3 + A74 * k4 + A75 * k5 + A76 * k6)
            k7 = f(xph, y1)
            e = (E1 * k1 + E3 * k3 + E4 * k4 + E5 * k5 + E6 * k6 + E7 * k7) * h
            self.nfcn += 6
            sk = self.atol + self.rtol * np.maximum(np.abs(y), np.abs(y1))
            err = np.sqrt(_seqsum((e / sk) ** 2) / n)
            fac11 = err ** expo1
            fac = fac11 / facold ** self.beta
            fac = max(facc2, min(facc1, fac / self.safe))
            hnew = h / fac
            self.log.append((x, h, err, err <= 1.0))
            if err <= 1.0:
                facold = max(err, 1.0e-4)
                naccpt += 1
                if naccpt % self.nstiff == 0 or iasti > 0:
                    stnum = _seqsum((k7 - k6) ** 2)
                    stden = _seqsum((y1 - ysti) ** 2)
                    if stden > 0.0:
                        hlamb = h * np.sqrt(stnum / stden)
                    if hlamb > 3.25:
                        nonsti = 0
                        iasti += 1
                        if iasti == 15:
                            self.idid = -4
                            break
                    else:
                        nonsti += 1
                        if nonsti == 6:
                            iasti = 0
                k1, y, x = k7, y1, xph
                if last:
                    h = hnew
                    self.idid = 1
                    break
                if abs(hnew) > hmax:
                    hnew = posneg * hmax
                if reject:
                    hnew = posneg * min(abs(hnew), abs(h))
                reject = False
            else:
                hnew = h / min(facc1, fac11 / self.safe)
                reject = True
                last = False
            h = hnew
        self.h, self.y, self.t = h, y, x
        return y
