/*
 * oracle/plant_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar f64 restatement of the builder-defined linear plant of BASELINE config 4: the lazy
 * "advance the plant to now" of OdePlant.updateState (gymwipe/plants/core.py:38-49) with the ODE
 * world replaced by n = round((now - last)/dt) applications of  x <- A x + B u.
 * PARITY UNPINNED: the reference's plant is an ODE rigid-body world (py3ode, absent) inside an env
 * that cannot be constructed; this file only pins the HIP kernel to the recurrence it claims to compute.
 */
#include <math.h>
#include <stdint.h>

/* advance one env; returns the number of substeps taken */
long long plant_oracle_update(const double* A, const double* B, double dt, double* x, double u,
                              double* t_last, double now)
{
    if (!(now > *t_last)) return 0;
    const long long n = llrint((now - *t_last) * (1.0 / dt));
    for (long long s = 0; s < n; ++s) {
        double y[4];
        for (int i = 0; i < 4; ++i) {
            double acc = B[i] * u;
            for (int j = 0; j < 4; ++j) acc += A[i * 4 + j] * x[j];
            y[i] = acc;
        }
        for (int i = 0; i < 4; ++i) x[i] = y[i];
    }
    if (n > 0) *t_last = now;
    return n;
}

void plant_oracle_update_batch(const double* A, const double* B, double dt, int64_t n_envs, double* x /*[N][4]*/,
                               const double* u, double* t_last, const double* now, uint64_t* substeps)
{
    for (int64_t e = 0; e < n_envs; ++e)
        substeps[e] += (uint64_t)plant_oracle_update(A, B, dt, x + e * 4, u[e], t_last + e, now[e]);
}
