// plant_mfma.hip -- batched linear plant  x <- A^n x + (sum_{j<n} A^j B) u  on the f64 matrix cores.
//
// Replaces OdePlant.updateState (gymwipe/plants/core.py:38-49: "step the plant by the simulated time
// that passed since the last update") for the builder-defined linear plant of BASELINE config 4.
//
// Mapping onto v_mfma_f64_16x16x4_f64 (D[16x16] = A[16x4] * B[4x16] + C):
//   K = 4   the state dimension,
//   N = 16  environments per instruction,
//   M = 16  FOUR candidate substep counts k0..k0+3 times 4 state components: row 4c+g of the A
//           operand is row g of A^(k0+c).
// Lane l = (g = l>>4, col = l&15) holds component g of env col as the B operand, and after the
// instruction holds component g of all four candidates in its four result registers
// (row = g + 4*reg).  So the result of one application is already where the next application needs
// it, and choosing the candidate that matches the env's substep count is a per-lane register select.
// The input term Q_k u (Q_k = sum_{j<k} A^j B) is rank one: ONE fused multiply-add per lane with the lane's own
// Q_k[g] -- as a second MFMA per candidate group it used one K-row of four and doubled the matrix-core time.
// Envs needing more than GW_PLANT_KMAX substeps go round the loop again.
#include <hip/hip_runtime.h>
#include "gw_internal.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));


namespace {

__global__ __launch_bounds__(64) void plant_update_kernel(GwPlantDev p, const char* __restrict__ now_base, int64_t stride,
                                                           int32_t* __restrict__ obs, float* __restrict__ reward,
                                                           double* __restrict__ angle_deg)
{
    const int lane = threadIdx.x;
    const int g = lane >> 4, col = lane & 15;
    const int64_t tiles = (p.N + 15) >> 4;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t e = tile * 16 + col;
        const bool live = e < p.N;
        double xg = 0.0, u = 0.0, now = 0.0, tl = 0.0;
        if (live) {
            xg = p.x[e * 4 + g];
            u = p.u[e];
            tl = p.t_last[e];
            now = *reinterpret_cast<const double*>(now_base + e * stride);
        }
        // substeps to take: n = round((now - last) / dt), nothing if time did not advance
        long long n = 0;
        if (live && now > tl) n = llrint((now - tl) * p.inv_dt);
        const long long n_total = n;
        while (__any(n > 0)) {
            const int chunk = n > GW_PLANT_KMAX ? GW_PLANT_KMAX : (int)n;   // this round's substeps (0 = done)
            const int mygrp = (chunk - 1) >> 2;                             // the candidate group holding it (-1: none)
            // every needed group's MFMA accumulates into the same registers; an env's state enters only the MFMA of its
            // own group (B operand zero elsewhere: exact), so one select among four candidates remains (ct_step_sfx.hip,
            // pend_step_kernel, performs the identical arithmetic for four tiles at once)
            v4f64 acc = {0.0, 0.0, 0.0, 0.0};
            for (int grp = 0; grp < GW_PLANT_KMAX / 4 && __any(chunk > 4 * grp); ++grp) {   // candidates k0..k0+3, k0 = 4*grp + 1
                const double a_p = p.Pop[grp * 64 + lane];
                const double b = (mygrp == grp) ? xg : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_p, b, acc, 0, 0, 0);
            }
            const int r = (chunk - 1) & 3;
            const double lo = (r & 1) ? acc.y : acc.x, hi = (r & 1) ? acc.w : acc.z;
            const double pick = (r & 2) ? hi : lo;
            xg = chunk > 0 ? fma(p.Qtab[chunk * 4 + g], u, pick) : xg;      // + Q_k u
            n -= chunk;
        }
        if (live && n_total > 0) {
            p.x[e * 4 + g] = xg;
            if (g == 0) { p.t_last[e] = now; p.nsub[e] += (unsigned long long)n_total; }
        }
        if (live && g == 2) {                                        // this lane holds the angle: interpreter feedback
            const double deg = xg * (180.0 / 3.141592653589793);     // (envs/inverted_pendulum.py:27-57), fused
            if (obs) obs[e] = (int32_t)deg;
            if (reward) reward[e] = (float)fabs(180.0 - deg);
            if (angle_deg) angle_deg[e] = deg;
        }
    }
}

__global__ void plant_set_input_kernel(GwPlantDev p, const double* __restrict__ u, const uint8_t* __restrict__ mask)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < p.N && (!mask || mask[e])) p.u[e] = u[e];
}

__global__ void plant_init_kernel(GwPlantDev p, double x0, double x1, double x2, double x3, double u0)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.N) return;
    p.x[e * 4 + 0] = x0; p.x[e * 4 + 1] = x1; p.x[e * 4 + 2] = x2; p.x[e * 4 + 3] = x3;
    p.u[e] = u0;
    p.t_last[e] = 0.0;
    p.nsub[e] = 0ull;
}

} // namespace

int gw_plant_launch_update(const GwPlantDev& p, const void* now_base, int64_t stride, int32_t* obs, float* reward, double* angle_deg,
                           void* stream)
{
    const int64_t tiles = (p.N + 15) >> 4;
    const unsigned grid = (unsigned)(tiles < 4096 ? tiles : 4096);   // >> 256 CUs; grid-stride over the rest
    hipLaunchKernelGGL(plant_update_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, p, (const char*)now_base, stride,
                       obs, reward, angle_deg);
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}

// InvertedPendulumInterpreter (envs/inverted_pendulum.py:27-57): observation int(degrees(angle)), reward
// float(abs(180 - degrees(angle))); math.degrees(x) is x * (180 / pi) in CPython
__global__ void plant_feedback_kernel(GwPlantDev p, int32_t* __restrict__ obs, float* __restrict__ reward, double* __restrict__ angle_deg)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.N) return;
    const double deg = p.x[e * 4 + 2] * (180.0 / 3.141592653589793);
    if (obs) obs[e] = (int32_t)deg;                       // int(): truncation towards zero
    if (reward) reward[e] = (float)fabs(180.0 - deg);
    if (angle_deg) angle_deg[e] = deg;
}

int gw_plant_launch_feedback(const GwPlantDev& p, int32_t* obs, float* reward, double* angle_deg, void* stream)
{
    hipLaunchKernelGGL(plant_feedback_kernel, dim3((unsigned)((p.N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, obs, reward, angle_deg);
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}

int gw_plant_launch_set_input(const GwPlantDev& p, const double* u, const uint8_t* mask, void* stream)
{
    hipLaunchKernelGGL(plant_set_input_kernel, dim3((unsigned)((p.N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, u, mask);
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}

int gw_plant_launch_init(const GwPlantDev& p, const double* x0, double u0, void* stream)
{
    hipLaunchKernelGGL(plant_init_kernel, dim3((unsigned)((p.N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p,
                       x0[0], x0[1], x0[2], x0[3], u0);
    return hipGetLastError() == hipSuccess ? GW_OK : GW_EHIP;
}
