// Continuous polish of an acquisition optimum: a bound-constrained limited-memory BFGS on the host around ONE device call per
// evaluation (alabi_gp_predict_grad_point: value and gradient of the GP mean and variance at a point).
//
// Reference: alabi/utility.py:1030-1163 (minimize_objective: scipy L-BFGS-B restarts on the acquisition function, one GP prediction
// per objective call) and :704-850 (grad_bape_utility / grad_agp_utility).  Here the optimiser runs next to the kernels: an
// active-learning iteration makes ~35 evaluations, and through scipy each cost 80 us at N = 100 (31 us of device work + 20 us of
// Python glue + 30 us inside scipy's wrappers) -- the polish was 70 % of the iteration.  In this file an evaluation is the device work
// plus a few hundred nanoseconds.
//
// Method: projected L-BFGS.  Variables on a bound whose gradient points outward are held; the two-loop recursion (memory 10) gives the
// direction on the others; a backtracking line search along the projected path x(t) = clip(x + t d) with the Armijo condition on the
// projected step (a failed search is repeated along the steepest descent with the memory dropped; steps without curvature information
// start at a tenth of the box and double while that pays); curvature pairs are kept when s.y > 1e-10 |s| |y|.  Stops on the projected-gradient norm (gtol), the relative
// decrease (ftol) or the iteration count.  The result is never worse than the start.
#include <hip/hip_runtime.h>
#include <cmath>
#include <vector>
#include "common.hpp"

namespace alabi {

// value and TRUE gradient of the acquisition function from (mu, var, dmu, dvar) by the chain rule (alabi_amd/utility.py:
// utility_value_and_grad; bape -[2 mu + var + log(e^var - 1)], agp -[mu + log(2 pi e var) / 2], jones = expected improvement)
static bool utility_value_grad(int algo, int d, const double* r /* mu, var, dmu[d], dvar[d] */, double y_best, double* u, double* g) {
    const double mu = r[0], var = r[1];
    const double* dmu = r + 2; const double* dvar = r + 2 + d;
    if (!(var > 0.0)) return false;
    double cm, cv;                                             // du = cm dmu + cv dvar
    if (algo == 0) {
        *u = -((2.0 * mu + var) + (var + std::log(1.0 - std::exp(-var))));       // logsubexp(var, 0) = var + log(1 - e^-var)
        cm = -2.0; cv = -(1.0 - 1.0 / std::expm1(-var));
    } else if (algo == 1) {
        *u = -(mu + 0.5 * std::log(2.0 * M_PI * M_E * var));
        cm = -1.0; cv = -0.5 / var;
    } else {
        const double sd = std::sqrt(var), z = (mu - y_best - 0.01) / sd;
        const double Phi = 0.5 * std::erfc(-z * M_SQRT1_2), phi = std::exp(-0.5 * z * z) / std::sqrt(2.0 * M_PI);
        *u = -((mu - y_best - 0.01) * Phi + sd * phi);
        cm = -Phi; cv = -phi / (2.0 * sd);
    }
    if (!std::isfinite(*u)) return false;
    for (int k = 0; k < d; ++k) {
        g[k] = cm * dmu[k] + cv * dvar[k];
        if (!std::isfinite(g[k])) return false;
    }
    return true;
}

}  // namespace alabi

using namespace alabi;

extern "C" int alabi_utility_polish(alabi_gp* gp, int algo, const double* x0, const double* bounds, double y_best, int maxiter,
                                    double* x_out, double* u_out, int* nevals_out, void* stream) {
    if (!gp || algo < 0 || algo > 2 || !x0 || !bounds || !x_out || !u_out || maxiter < 0) return ALABI_BAD_ARGUMENT;
    if (!gp->computed || !gp->has_alpha) return ALABI_NOT_COMPUTED;
    const int d = gp->d, M = 10;
    std::vector<double> lo(d), hi(d), x(d), g(d), xn(d), gn(d), dir(d), res(2 + 2 * d), q(d), alpha(M), rho(M);
    std::vector<std::vector<double>> S(M, std::vector<double>(d)), Y(M, std::vector<double>(d));
    for (int k = 0; k < d; ++k) {
        // the box shrunk by 1e-9 of its width: the reference's objective is +inf ON the boundary (utility.py:268-275)
        const double e = 1e-9 * (bounds[2 * k + 1] - bounds[2 * k]);
        lo[k] = bounds[2 * k] + e; hi[k] = bounds[2 * k + 1] - e;
        x[k] = std::fmin(std::fmax(x0[k], lo[k]), hi[k]);
    }
    int nevals = 0, st = ALABI_OK;
    auto eval = [&](const std::vector<double>& p, double* u, std::vector<double>& grad) -> bool {   // false: not finite there
        ++nevals;
        st = alabi_gp_predict_grad_point(gp, p.data(), res.data(), stream);
        if (st != ALABI_OK) return false;
        return utility_value_grad(algo, d, res.data(), y_best, u, grad.data());
    };
    double f;
    const bool ok0 = eval(x, &f, g);
    if (st != ALABI_OK) return st;
    for (int k = 0; k < d; ++k) x_out[k] = x[k];
    *u_out = ok0 ? f : INFINITY;
    if (nevals_out) *nevals_out = nevals;
    if (!ok0) return ALABI_OK;                                  // nothing to descend from
    int npairs = 0, head = 0;                                  // ring of the last M curvature pairs
    const double ftol = 1e-12, gtol = 1e-8;
    for (int it = 0; it < maxiter; ++it) {
        // held variables: on a bound with the gradient pointing outward
        double pg = 0.0;
        for (int k = 0; k < d; ++k) {
            const bool held = (x[k] <= lo[k] && g[k] > 0.0) || (x[k] >= hi[k] && g[k] < 0.0);
            q[k] = held ? 0.0 : g[k];
            pg = std::fmax(pg, std::fabs(q[k]));
        }
        if (pg < gtol) break;
        double fn = f;
        bool accepted = false;
        for (int attempt = 0; attempt < 2 && !accepted; ++attempt) {
            // two-loop recursion on the free part (second attempt, after a failed search: steepest descent, memory dropped)
            if (attempt == 1) { if (npairs == 0) break; npairs = 0; }
            for (int k = 0; k < d; ++k) dir[k] = q[k];
            for (int j = 0; j < npairs; ++j) {
                const int i = (head - 1 - j + 2 * M) % M;
                double a = 0.0;
                for (int k = 0; k < d; ++k) a += S[i][k] * dir[k];
                alpha[i] = a * rho[i];
                for (int k = 0; k < d; ++k) dir[k] -= alpha[i] * Y[i][k];
            }
            if (npairs > 0) {
                const int i = (head - 1 + M) % M;
                double yy = 0.0, sy = 0.0;
                for (int k = 0; k < d; ++k) { yy += Y[i][k] * Y[i][k]; sy += S[i][k] * Y[i][k]; }
                const double gamma = sy / yy;
                for (int k = 0; k < d; ++k) dir[k] *= gamma;
            }
            for (int j = npairs - 1; j >= 0; --j) {
                const int i = (head - 1 - j + 2 * M) % M;
                double b = 0.0;
                for (int k = 0; k < d; ++k) b += Y[i][k] * dir[k];
                b *= rho[i];
                for (int k = 0; k < d; ++k) dir[k] += (alpha[i] - b) * S[i][k];
            }
            double slope = 0.0;
            for (int k = 0; k < d; ++k) { dir[k] = (q[k] == 0.0 && g[k] != 0.0) ? 0.0 : -dir[k]; slope += g[k] * dir[k]; }
            if (!(slope < 0.0)) {                               // not a descent direction: steepest descent on the free part
                for (int k = 0; k < d; ++k) dir[k] = -q[k];
                npairs = 0;
            }
            double t = 1.0;
            if (npairs == 0) {                                  // no curvature yet: a first step of a tenth of the box at most
                double scale = 0.0;
                for (int k = 0; k < d; ++k) scale = std::fmax(scale, std::fabs(dir[k]) / (hi[k] - lo[k]));
                t = scale > 0.0 ? std::fmin(1.0, 0.1 / scale) : 1.0;
            }
            // backtracking along the projected path; without curvature information an accepted first trial is followed by doublings
            // while they pay
            auto trial = [&](double tt, std::vector<double>& xp, std::vector<double>& gp_, double* fp, double* decr) -> int {   // 1 accepted, 0 rejected, -1 no move
                double dd = 0.0, moved = 0.0;
                for (int k = 0; k < d; ++k) {
                    xp[k] = std::fmin(std::fmax(x[k] + tt * dir[k], lo[k]), hi[k]);
                    dd += g[k] * (xp[k] - x[k]);
                    moved = std::fmax(moved, std::fabs(xp[k] - x[k]));
                }
                *decr = dd;
                if (moved == 0.0) return -1;
                const bool fin = eval(xp, fp, gp_);
                return (fin && *fp <= f + 1e-4 * dd) ? 1 : 0;
            };
            for (int ls = 0; ls < 12; ++ls) {
                double decr;
                const int r = trial(t, xn, gn, &fn, &decr);
                if (st != ALABI_OK) return st;
                if (r < 0) break;
                if (r == 0) {
                    // next trial: the minimiser of the parabola through f, its slope along the projected step and the rejected value, kept
                    // inside [0.1 t, 0.5 t]; a quarter of the step where the value was not finite
                    double tn = 0.25 * t;
                    if (std::isfinite(fn) && decr < 0.0) {
                        const double curv = fn - f - decr;             // = (1/2) phi'' t^2 for a parabola along the step
                        if (curv > 0.0) tn = std::fmin(std::fmax(-0.5 * decr / curv * t, 0.1 * t), 0.5 * t);
                    }
                    t = tn;
                    continue;
                }
                if (r == 1) {
                    accepted = true;
                    if (ls == 0 && npairs == 0) {           // (a quasi-Newton step is taken as it is)
                        std::vector<double> x2(d), g2(d);
                        for (int e = 0; e < 4; ++e) {
                            double f2, d2;
                            const int r2 = trial(2.0 * t, x2, g2, &f2, &d2);
                            if (st != ALABI_OK) return st;
                            if (r2 != 1 || !(f2 < fn)) break;
                            xn = x2; gn = g2; fn = f2; t *= 2.0;
                        }
                    }
                    break;
                }
            }
        }
        if (!accepted) break;
        double sy = 0.0, ss = 0.0, yy = 0.0;
        for (int k = 0; k < d; ++k) {
            const double s = xn[k] - x[k], yv = gn[k] - g[k];
            S[head][k] = s; Y[head][k] = yv;
            sy += s * yv; ss += s * s; yy += yv * yv;
        }
        if (sy > 1e-10 * std::sqrt(ss * yy)) { rho[head] = 1.0 / sy; head = (head + 1) % M; if (npairs < M) ++npairs; }
        const double df = f - fn;
        x = xn; g = gn; f = fn;
        if (df <= ftol * std::fmax(std::fmax(std::fabs(f), std::fabs(f + df)), 1.0)) break;
    }
    if (f < *u_out) {
        for (int k = 0; k < d; ++k) x_out[k] = x[k];
        *u_out = f;
    }
    if (nevals_out) *nevals_out = nevals;
    return ALABI_OK;
}
