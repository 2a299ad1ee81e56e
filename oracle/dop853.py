"""TEST INFRASTRUCTURE -- CPU restatement of the integrator behind the reference's calc='dop853' method-of-lines path.

The reference (catint/calculator_old.py:955-963) builds scipy.integrate.ode(ode_func).set_integrator('dop853', nsteps=10000) and calls
r.integrate(r.t + dt) once per output interval.  scipy's 'dop853' wraps E. Hairer's DOP853 (Hairer, Norsett, Wanner, Solving ODEs I,
2nd ed., II.5; dop853.f): the 12-stage Dormand-Prince method of order 8 with embedded 5th- and 3rd-order error estimators, the same
step-size controller, HINIT and stiffness detection as DOPRI5 (oracle/dopri5.py) with the order-8 constants.  Third-party dependency
of the reference (scipy, not vendored; this image: scipy 1.15.3); the coefficients are Hairer's, taken here from the table scipy ships
for its own Python port (scipy.integrate._ivp.dop853_coefficients: A, B, C, E5; the 3rd-order estimator through bhh1..3).  Pinned
against scipy.integrate.ode('dop853') itself in tests/test_ode_oracle.py (same evaluation times, same trajectory bit for bit).

Wrapper behaviour as in oracle/dopri5.py: a fresh DOP853 call per integrate(), step size carried in WORK(7).  scipy's defaults:
rtol 1e-6, atol 1e-12, safety 0.9, dfactor 0.3 (scipy's own choice; FAC1), ifactor 6 (FAC2), beta 0 (DOP853: no stabilisation),
NSTIFF 1000, UROUND 2.3e-16.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import numpy as np
from scipy.integrate._ivp import dop853_coefficients as _dc

from .dopri5 import UROUND, _seqsum

A = _dc.A          # A[s][j]: coefficient of k_{j+1} in the argument of stage s+1
B = _dc.B
C = _dc.C
ER = _dc.E5        # er1, er6 ... er12
BHH1 = 0.244094488188976377952755905512e+00
BHH2 = 0.733846688281611857341361741547e+00
BHH3 = 0.220588235294117647058823529412e-01
# the nonzero columns of every stage row, in the order the Fortran source adds them
STAGE_COLS = {1: [0], 2: [0, 1], 3: [0, 2], 4: [0, 2, 3], 5: [0, 3, 4], 6: [0, 3, 4, 5], 7: [0, 3, 4, 5, 6], 8: [0, 3, 4, 5, 6, 7],
              9: [0, 3, 4, 5, 6, 7, 8], 10: [0, 3, 4, 5, 6, 7, 8, 9], 11: [0, 3, 4, 5, 6, 7, 8, 9, 10]}
B_COLS = [0, 5, 6, 7, 8, 9, 10, 11]


class Dop853(object):
    def __init__(self, f, rtol=1e-6, atol=1e-12, nsteps=500, max_step=0.0, first_step=0.0, safety=0.9, ifactor=6.0, dfactor=0.3,
                 beta=0.0, nstiff=1000, recompute_k1=True):
        self.f = f
        # scipy's build of dop853.f evaluates f(x, y) again at the start of every step (the `IF (IRTRN.GE.2) CALL FCN` branch is taken
        # although no dense output is requested): same operands, same k1, one more evaluation per attempted step.  Kept here so
        # that the evaluation sequence can be pinned; the device integrator leaves it out (recompute_k1=False gives its counts).
        self.recompute_k1 = bool(recompute_k1)
        self.rtol, self.atol, self.nmax = float(rtol), float(atol), int(nsteps)
        self.max_step, self.h = float(max_step), float(first_step)
        self.safe, self.fac1, self.fac2 = float(safety), float(dfactor), float(ifactor)
        self.beta = 0.0 if beta <= 0.0 else float(beta)
        self.nstiff = int(nstiff)
        self.idid = 1
        self.log = []
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
        h1 = max(1.0e-6, abs(h) * 1.0e-3) if der12 <= 1e-15 else (0.01 / der12) ** (1.0 / 8.0)
        return min(100 * abs(h), h1, hmax) * posneg

    def integrate(self, xend):
        f, y, x = self.f, self.y, self.t
        n = y.size
        hmax = abs(self.max_step if self.max_step != 0.0 else xend - x)
        facold, expo1 = 1.0e-4, 1.0 / 8.0 - self.beta * 0.2
        facc1, facc2 = 1.0 / self.fac1, 1.0 / self.fac2
        posneg = 1.0 if xend - x >= 0 else -1.0
        last, reject = False, False
        hlamb, iasti, nonsti = 0.0, 0, 0
        nstep = naccpt = 0
        k = [None] * 12
        k[0] = f(x, y)
        h = self.h
        if h == 0.0:
            h = self._hinit(x, y, posneg, k[0], hmax)
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
            if self.recompute_k1:
                k[0] = f(x, y)
                self.nfcn_quirk = getattr(self, 'nfcn_quirk', 0) + 1
            y1 = y + h * A[1][0] * k[0]
            k[1] = f(x + C[1] * h, y1)
            for s in range(2, 12):
                cols = STAGE_COLS[s]
                acc = A[s][cols[0]] * k[cols[0]]
                for j in cols[1:]:
                    acc = acc + A[s][j] * k[j]
                y1 = y + h * acc
                k[s] = f(x + (C[s] * h if s < 11 else h), y1) if s < 11 else f(x + h, y1)
            xph = x + h
            acc = B[0] * k[0]
            for j in B_COLS[1:]:
                acc = acc + B[j] * k[j]
            k4 = acc
            k5 = y + h * k4
            self.nfcn += 11
            sk = self.atol + self.rtol * np.maximum(np.abs(y), np.abs(k5))
            erri = k4 - BHH1 * k[0] - BHH2 * k[8] - BHH3 * k[11]
            err2 = _seqsum((erri / sk) ** 2)
            erri = ER[0] * k[0]
            for j in B_COLS[1:]:
                erri = erri + ER[j] * k[j]
            err = _seqsum((erri / sk) ** 2)
            deno = err + 0.01 * err2
            if deno <= 0.0:
                deno = 1.0
            err = abs(h) * err * np.sqrt(1.0 / (n * deno))
            fac11 = err ** expo1
            fac = fac11 / facold ** self.beta
            fac = max(facc2, min(facc1, fac / self.safe))
            hnew = h / fac
            self.log.append((x, h, err, err <= 1.0))
            if err <= 1.0:
                facold = max(err, 1.0e-4)
                naccpt += 1
                knew = f(xph, k5)
                self.nfcn += 1
                if naccpt % self.nstiff == 0 or iasti > 0:
                    stnum = _seqsum((knew - k[11]) ** 2)
                    stden = _seqsum((k5 - y1) ** 2)
                    if stden > 0.0:
                        hlamb = abs(h) * np.sqrt(stnum / stden)
                    if hlamb > 6.1:
                        nonsti = 0
                        iasti += 1
                        if iasti == 15:
                            self.idid = -4
                            break
                    else:
                        nonsti += 1
                        if nonsti == 6:
                            iasti = 0
                k[0], y, x = knew, k5, xph
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
