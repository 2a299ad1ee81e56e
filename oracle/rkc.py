"""TEST INFRASTRUCTURE (oracle): Runge-Kutta-Chebyshev of second order with error control, the algorithm of
B. P. Sommeijer, L. F. Shampine, J. G. Verwer, "RKC: an explicit solver for parabolic PDEs", J. Comput. Appl. Math. 88 (1998) 315-326
(their Fortran code `rkc.f`: stages by the three-term Chebyshev recurrence with damping 2/13, error estimate
0.8 (y_n - y_{n+1}) + 0.4 h (f_n + f_{n+1}), step-size controller with memory, spectral radius by a nonlinear power iteration),
restated from the paper.  It is what pnp_integrate_rkc (catint_amd/csrc/pnp_rkc.hip) runs per lane on the device: the batched,
stiffness-capable counterpart of the reference's scipy.integrate.odeint / ode('vode' | 'lsoda') drivers of the method of lines
(catint/calculator_old.py:946-963), which integrate ONE operating point per call on the host.

Parity: there is no RKC in the reference, so nothing here is pinned to the reference's output bit for bit; the device integrator is
checked against this restatement (same steps, same stage counts) and both against scipy's `odeint` on the reference's own right-hand
side within the requested tolerance (tests/test_ode_oracle.py, tests/test_gpu_ode.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import math

import numpy as np

UROUND = 2.22e-16


class Rkc(object):
    """One integrator instance = one operating point.  integrate(tend) advances from self.t to tend, carrying step size, spectral
    radius and eigenvector estimate from call to call (the device driver does one call per output interval)."""

    def __init__(self, f, rtol=1e-6, atol=1e-12, nsteps=100000, max_step=0.0):
        self.f = f
        self.rtol, self.atol = float(rtol), float(atol)
        self.nmax = int(nsteps)
        self.max_step = float(max_step)
        self.mmax = max(int(round(math.sqrt(self.rtol / (10.0 * UROUND)))), 2)
        self.nfe = self.nfesig = self.nsteps = self.naccpt = self.nrejct = self.maxm = 0
        self.idid = 1
        self.log = []          # (t, h, m, err, accepted)

    def set_initial_value(self, y, t=0.0):
        self.y = np.array(y, dtype=float)
        self.t = float(t)
        self.started = False
        return self

    def successful(self):
        return self.idid == 1

    # -- spectral radius: nonlinear power iteration on f around yn -----------------------------------------------------------------
    def _rho(self, hmax):
        yn, fn = self.y, self.fn
        v = self.ev if self.have_ev else fn
        ynrm = math.sqrt(float(np.sum(yn * yn)))
        vnrm = math.sqrt(float(np.sum(v * v)))
        sq = math.sqrt(UROUND)
        if ynrm != 0.0 and vnrm != 0.0:
            dynrm = ynrm * sq
            v = yn + v * (dynrm / vnrm)
        elif ynrm != 0.0:
            dynrm = ynrm * sq
            v = yn + yn * sq
        elif vnrm != 0.0:
            dynrm = UROUND
            v = v * (dynrm / vnrm)
        else:
            dynrm = UROUND
            v = np.full_like(yn, dynrm)
        sigma = 0.0
        for it in range(1, 51):
            fv = self.f(self.t, v)
            self.nfesig += 1
            d = fv - fn
            dfnrm = math.sqrt(float(np.sum(d * d)))
            sigmal = sigma
            sigma = dfnrm / dynrm
            self.sprad = 1.2 * sigma
            if it >= 2 and abs(sigma - sigmal) <= max(sigma, 1.0 / hmax) * 0.01:
                self.ev = v - yn
                self.have_ev = True
                return True
            if dfnrm != 0.0:
                v = yn + d * (dynrm / dfnrm)
            else:
                # f is locally constant in this direction: another direction (rkc.f flips one component; here: the state itself)
                v = yn + yn * sq if ynrm != 0.0 else np.full_like(yn, dynrm)
        return False

    def integrate(self, tend):
        if self.idid != 1:
            return self.y
        f = self.f
        tend = float(tend)
        n = self.y.size
        hmax = abs(tend - self.t) if self.max_step == 0.0 else min(self.max_step, abs(tend - self.t))
        if not self.started:
            self.fn = f(self.t, self.y)
            self.nfe += 1
            self.have_ev = False
            self.newspc = True
            self.jacatt = False
            self.nstsig = 0
            self.errold = 0.0
            self.hold = 0.0
            self.absh = 0.0
            self.first = True
            self.started = True
        nstep_call = 0
        last = False
        while True:
            hmin = 10.0 * UROUND * max(abs(self.t), hmax)
            if self.newspc:
                if not self._rho(hmax):
                    self.idid = -6
                    return self.y
                self.jacatt = True
                self.newspc = False
            if self.first:
                # initial step size: 1/sprad, corrected by a first-order estimate of the local error of an Euler step
                absh = hmax
                if self.sprad * absh > 1.0:
                    absh = 1.0 / self.sprad
                absh = max(absh, hmin)
                v = self.y + absh * self.fn
                fv = f(self.t + absh, v)
                self.nfe += 1
                wt = self.atol + self.rtol * np.abs(self.y)
                est = absh * math.sqrt(float(np.sum(((fv - self.fn) / wt) ** 2)) / n)
                if 0.1 * absh < hmax * math.sqrt(est):
                    absh = max(0.1 * absh / math.sqrt(est), hmin)
                else:
                    absh = hmax
                self.absh = absh
                self.first = False
            absh = min(self.absh, hmax)
            last = False
            if 1.1 * absh >= abs(tend - self.t):
                absh = abs(tend - self.t)
                last = True
            m = 1 + int(math.sqrt(1.54 * absh * self.sprad + 1.0))
            if m > self.mmax:
                m = self.mmax
                absh = (m * m - 1) / (1.54 * self.sprad)
                last = False
            self.maxm = max(self.maxm, m)
            h = absh
            nstep_call += 1
            if nstep_call > self.nmax:
                self.idid = -2
                return self.y
            # ---- one step: m stages ----
            yn, fn = self.y, self.fn
            w0 = 1.0 + 2.0 / (13.0 * m * m)
            t1 = w0 * w0 - 1.0
            t2 = math.sqrt(t1)
            arg = m * math.log(w0 + t2)
            w1 = math.sinh(arg) * t1 / (math.cosh(arg) * m * t2 - w0 * math.sinh(arg))
            bjm1 = bjm2 = 1.0 / (2.0 * w0) ** 2
            yjm2 = yn
            mus = w1 * bjm1
            yjm1 = yn + (h * mus) * fn
            zjm1, zjm2, dzjm1, dzjm2, d2zjm1, d2zjm2 = w0, 1.0, 1.0, 0.0, 0.0, 0.0
            for j in range(2, m + 1):
                zj = 2.0 * w0 * zjm1 - zjm2
                dzj = 2.0 * w0 * dzjm1 - dzjm2 + 2.0 * zjm1
                d2zj = 2.0 * w0 * d2zjm1 - d2zjm2 + 4.0 * dzjm1
                bj = d2zj / (dzj * dzj)
                ajm1 = 1.0 - zjm1 * bjm1
                mu = 2.0 * w0 * bj / bjm1
                nu = -bj / bjm2
                mus = mu * w1 / w0
                fj = f(self.t, yjm1)
                self.nfe += 1
                y = mu * yjm1 + nu * yjm2 + (1.0 - mu - nu) * yn + (h * mus) * (fj - ajm1 * fn)
                yjm2, yjm1 = yjm1, y
                bjm2, bjm1 = bjm1, bj
                zjm2, zjm1 = zjm1, zj
                dzjm2, dzjm1 = dzjm1, dzj
                d2zjm2, d2zjm1 = d2zjm1, d2zj
            ynew = yjm1
            fnew = f(self.t + h, ynew)
            self.nfe += 1
            self.nsteps += 1
            wt = self.atol + self.rtol * np.maximum(np.abs(ynew), np.abs(yn))
            est = 0.8 * (yn - ynew) + 0.4 * h * (fn + fnew)
            err = math.sqrt(float(np.sum((est / wt) ** 2)) / n)
            if not (err <= 1.0):          # (NaN rejects as well)
                self.log.append((self.t, h, m, err, False))
                self.nrejct += 1
                if not math.isfinite(err):
                    absh = 0.1 * absh
                else:
                    absh = 0.8 * absh / err ** (1.0 / 3.0)
                if absh < hmin:
                    self.idid = -3
                    return self.y
                self.absh = absh
                self.newspc = not self.jacatt
                continue
            # accepted
            self.log.append((self.t, h, m, err, True))
            self.naccpt += 1
            self.t = tend if last else self.t + h
            self.jacatt = False
            self.nstsig = (self.nstsig + 1) % 25
            self.newspc = self.nstsig == 0
            self.y, self.fn = ynew, fnew
            fac = 10.0
            if self.naccpt == 1:
                t2 = err ** (1.0 / 3.0)
                if 0.8 < fac * t2:
                    fac = 0.8 / t2
            else:
                t1 = 0.8 * absh * self.errold ** (1.0 / 3.0)
                t2 = abs(self.hold) * err ** (2.0 / 3.0)
                if t1 < fac * t2:
                    fac = t1 / t2
            self.absh = max(hmin, max(0.1, fac) * absh)
            self.errold = err
            self.hold = h
            if last:
                return self.y
