// gw_plant_api.cpp -- C-ABI of the linear plant (include/gymwipe_amd.h, "Linear plant"): host side.
#include "gw_internal.h"

#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <vector>


int gw_plant_launch_update(const GwPlantDev& p, const void* now_base, int64_t stride, int32_t* obs, float* reward, double* angle_deg,
                           void* stream);
int gw_plant_launch_set_input(const GwPlantDev& p, const double* u, const uint8_t* mask, void* stream);
int gw_plant_launch_init(const GwPlantDev& p, const double* x0, double u0, void* stream);
int gw_plant_launch_feedback(const GwPlantDev& p, int32_t* obs, float* reward, double* angle_deg, void* stream);

int gw_set_error(int code, const char* fmt, ...);      // gw_api.cpp
int gw_env_internals(gw_env* env, const GwState** st, const GwDevConst** cst, int* hip_device);   // gw_api.cpp
void gw_env_add_steps(gw_env* env, uint64_t n);                                                    // gw_api.cpp
int gw_launch_pend_step(const GwState& st, const GwDevConst& cst, const GwPlantDev& p, const int32_t* device, const int32_t* duration,
                        int32_t* obs, float* reward, double* angle_deg, void* stream, bool below_limits);  // ct_step_sfx.hip
bool gw_env_below_limits(gw_env* env, void* stream);                                                      // gw_api.cpp

struct gw_plant {
    gw_plant_config cfg;
    GwPlantDev dev;
    void* blocks[8];
    int nblocks;
};

namespace {

#define PLANT_HIP(expr, cleanup)                                                                  \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            cleanup;                                                                              \
            return gw_set_error(GW_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e));          \
        }                                                                                         \
    } while (0)

template <class T>
int palloc(gw_plant* p, T** out, size_t count)
{
    void* q = nullptr;
    if (hipMalloc(&q, count * sizeof(T)) != hipSuccess) return gw_set_error(GW_ENOMEM, "hipMalloc failed in gw_plant_create");
    p->blocks[p->nblocks++] = q;
    *out = (T*)q;
    return GW_OK;
}

} // namespace

extern "C" {

int gw_plant_config_default(gw_plant_config* c, int64_t num_envs)
{
    if (!c) return gw_set_error(GW_EINVAL, "cfg is NULL");
    memset(c, 0, sizeof *c);
    c->abi_version = GW_ABI_VERSION;
    c->num_envs = num_envs;
    c->dt = 1e-3;                                   // the sensor's 1 ms sampling (envs/inverted_pendulum.py:81)
    // continuous model (builder-defined): wagon under a velocity servo, small-angle pendulum of arm 1 m
    //   p' = v;  v' = (u - v)/tau;  th' = w;  w' = -(g/l) th - c w - (1/l) v'
    // forward Euler at dt:  A = I + dt*Ac,  B = dt*Bc
    const double tau = 0.05, g_l = 9.81, damp = 0.2, dt = c->dt;
    const double Ac[16] = {0, 1, 0, 0,
                           0, -1 / tau, 0, 0,
                           0, 0, 0, 1,
                           0, 1 / tau, -g_l, -damp};
    const double Bc[4] = {0, 1 / tau, 0, -1 / tau};
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) c->A[i * 4 + j] = (i == j ? 1.0 : 0.0) + dt * Ac[i * 4 + j];
        c->B[i] = dt * Bc[i];
    }
    c->x0[0] = 0.0; c->x0[1] = 0.0; c->x0[2] = 0.05; c->x0[3] = 0.0;
    c->u0 = 0.1;                                    // slider.setParam(ParamVel, 0.1), sliding_pendulum.py:52
    return GW_OK;
}

int gw_plant_create(const gw_plant_config* cfg, gw_plant** out)
{
    if (!cfg || !out) return gw_set_error(GW_EINVAL, "cfg/out is NULL");
    *out = nullptr;
    if (cfg->abi_version != GW_ABI_VERSION) return gw_set_error(GW_EINVAL, "abi_version mismatch");
    if (cfg->num_envs <= 0 || !(cfg->dt > 0)) return gw_set_error(GW_EINVAL, "num_envs and dt must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return gw_set_error(GW_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
    if (cfg->hip_device < 0 || cfg->hip_device >= ndev) return gw_set_error(GW_EINVAL, "hip_device out of range");
    gw_plant* p = new (std::nothrow) gw_plant();
    if (!p) return gw_set_error(GW_ENOMEM, "out of host memory");
    memset(p, 0, sizeof *p);
    p->cfg = *cfg;
    PLANT_HIP(hipSetDevice(cfg->hip_device), delete p);

    // powers of A and the accumulated input vectors: P_1 = A, P_{k+1} = A P_k; q_1 = B, q_{k+1} = A q_k + B
    const int K = GW_PLANT_KMAX;
    std::vector<double> P((size_t)(K + 1) * 16), q((size_t)(K + 1) * 4);
    memcpy(&P[16], cfg->A, 16 * sizeof(double));
    memcpy(&q[4], cfg->B, 4 * sizeof(double));
    for (int k = 1; k < K; ++k)
        for (int i = 0; i < 4; ++i) {
            for (int j = 0; j < 4; ++j) {
                double s = 0;
                for (int m = 0; m < 4; ++m) s += cfg->A[i * 4 + m] * P[(size_t)k * 16 + m * 4 + j];
                P[(size_t)(k + 1) * 16 + i * 4 + j] = s;
            }
            double s = cfg->B[i];
            for (int m = 0; m < 4; ++m) s += cfg->A[i * 4 + m] * q[(size_t)k * 4 + m];
            q[(size_t)(k + 1) * 4 + i] = s;
        }
    // MFMA A operands by lane: lane l = (row = l & 15, kk = l >> 4); row = 4*candidate + component
    std::vector<double> pop((size_t)(K / 4) * 64), qop((size_t)(K + 1) * 4, 0.0);      // qop: Q_k[comp], row 0 = no substep
    for (int grp = 0; grp < K / 4; ++grp)
        for (int l = 0; l < 64; ++l) {
            const int row = l & 15, kk = l >> 4, cand = row >> 2, comp = row & 3, k = 4 * grp + 1 + cand;
            pop[(size_t)grp * 64 + l] = P[(size_t)k * 16 + comp * 4 + kk];
        }

    for (int k = 1; k <= K; ++k) for (int i = 0; i < 4; ++i) qop[(size_t)k * 4 + i] = q[(size_t)k * 4 + i];
    const int64_t N = cfg->num_envs;
    double *dP = nullptr, *dQ = nullptr;
    int rc;
#define PA(ptr, cnt) do { rc = palloc(p, &(ptr), (size_t)(cnt)); if (rc) { gw_plant_destroy(p); return rc; } } while (0)
    PA(p->dev.x, N * 4); PA(p->dev.u, N); PA(p->dev.t_last, N); PA(p->dev.nsub, N);
    PA(dP, pop.size()); PA(dQ, qop.size());
#undef PA
    p->dev.N = N; p->dev.Pop = dP; p->dev.Qtab = dQ; p->dev.dt = cfg->dt; p->dev.inv_dt = 1.0 / cfg->dt;
    PLANT_HIP(hipMemcpy(dP, pop.data(), pop.size() * sizeof(double), hipMemcpyHostToDevice), gw_plant_destroy(p));
    PLANT_HIP(hipMemcpy(dQ, qop.data(), qop.size() * sizeof(double), hipMemcpyHostToDevice), gw_plant_destroy(p));
    if (gw_plant_launch_init(p->dev, cfg->x0, cfg->u0, nullptr)) { gw_plant_destroy(p); return gw_set_error(GW_EHIP, "plant init launch failed"); }
    PLANT_HIP(hipDeviceSynchronize(), gw_plant_destroy(p));
    *out = p;
    return GW_OK;
}

int gw_plant_destroy(gw_plant* p)
{
    if (!p) return GW_OK;
    (void)hipSetDevice(p->cfg.hip_device);
    for (int i = 0; i < p->nblocks; ++i) (void)hipFree(p->blocks[i]);
    delete p;
    return GW_OK;
}

int gw_plant_update(gw_plant* p, const void* now_dev, int64_t stride_bytes, void* stream)
{
    if (!p || !now_dev) return gw_set_error(GW_EINVAL, "plant/now is NULL");
    if (stride_bytes < 8 || (stride_bytes & 7)) return gw_set_error(GW_EINVAL, "stride_bytes must be a multiple of 8");
    PLANT_HIP(hipSetDevice(p->cfg.hip_device), (void)0);
    if (gw_plant_launch_update(p->dev, now_dev, stride_bytes, nullptr, nullptr, nullptr, stream)) return gw_set_error(GW_EHIP, "plant update launch failed");
    return GW_OK;
}

int gw_plant_update_feedback(gw_plant* p, const void* now_dev, int64_t stride_bytes, int32_t* obs_dev, float* reward_dev,
                             double* angle_deg_dev, void* stream)
{
    if (!p || !now_dev) return gw_set_error(GW_EINVAL, "plant/now is NULL");
    if (stride_bytes < 8 || (stride_bytes & 7)) return gw_set_error(GW_EINVAL, "stride_bytes must be a multiple of 8");
    PLANT_HIP(hipSetDevice(p->cfg.hip_device), (void)0);
    if (gw_plant_launch_update(p->dev, now_dev, stride_bytes, obs_dev, reward_dev, angle_deg_dev, stream))
        return gw_set_error(GW_EHIP, "plant update launch failed");
    return GW_OK;
}

int gw_pendulum_step(gw_env* env, gw_plant* p, const int32_t* device_dev, const int32_t* duration_dev, int32_t* obs_dev,
                     float* reward_dev, double* angle_deg_dev, void* stream)
{
    if (!env || !p) return gw_set_error(GW_EINVAL, "env/plant is NULL");
    if (!device_dev || !duration_dev) return gw_set_error(GW_EINVAL, "gw_pendulum_step: NULL action pointer");
    const GwState* st = nullptr; const GwDevConst* cst = nullptr; int dev = 0;
    int rc = gw_env_internals(env, &st, &cst, &dev);
    if (rc) return rc;
    if (!st->tk) return gw_set_error(GW_EUNSUPPORTED, "gw_pendulum_step needs the default (suffix) queue mode");
    if (st->D != 2) return gw_set_error(GW_EUNSUPPORTED, "gw_pendulum_step: the env's network has two assignable devices (sensor, controller)");
    if (st->N != p->dev.N || dev != p->cfg.hip_device) return gw_set_error(GW_EINVAL, "env and plant differ in num_envs or hip_device");
    PLANT_HIP(hipSetDevice(dev), (void)0);
    if (gw_launch_pend_step(*st, *cst, p->dev, device_dev, duration_dev, obs_dev, reward_dev, angle_deg_dev, stream, gw_env_below_limits(env, stream)))
        return gw_set_error(GW_EHIP, "pendulum step launch failed");
    gw_env_add_steps(env, 1);
    return GW_OK;
}

int gw_plant_set_input(gw_plant* p, const double* u_dev, const uint8_t* mask_dev, void* stream)
{
    if (!p || !u_dev) return gw_set_error(GW_EINVAL, "plant/u is NULL");
    PLANT_HIP(hipSetDevice(p->cfg.hip_device), (void)0);
    if (gw_plant_launch_set_input(p->dev, u_dev, mask_dev, stream)) return gw_set_error(GW_EHIP, "plant set_input launch failed");
    return GW_OK;
}

int gw_plant_feedback(gw_plant* p, int32_t* obs_dev, float* reward_dev, double* angle_deg_dev, void* stream)
{
    if (!p) return gw_set_error(GW_EINVAL, "plant is NULL");
    PLANT_HIP(hipSetDevice(p->cfg.hip_device), (void)0);
    if (gw_plant_launch_feedback(p->dev, obs_dev, reward_dev, angle_deg_dev, stream)) return gw_set_error(GW_EHIP, "plant feedback launch failed");
    return GW_OK;
}

int gw_plant_state_ptr(gw_plant* p, double** x_dev)
{
    if (!p || !x_dev) return gw_set_error(GW_EINVAL, "plant/x is NULL");
    *x_dev = p->dev.x;
    return GW_OK;
}

int gw_plant_get_state(gw_plant* p, const char* field, void* dst, size_t bytes)
{
    if (!p || !field || !dst) return gw_set_error(GW_EINVAL, "plant/field/dst is NULL");
    PLANT_HIP(hipSetDevice(p->cfg.hip_device), (void)0);
    PLANT_HIP(hipDeviceSynchronize(), (void)0);
    const int64_t N = p->dev.N;
    const void* src = nullptr; size_t need = 0;
    if (!strcmp(field, "x")) { src = p->dev.x; need = (size_t)N * 4 * sizeof(double); }
    else if (!strcmp(field, "u")) { src = p->dev.u; need = (size_t)N * sizeof(double); }
    else if (!strcmp(field, "t_last")) { src = p->dev.t_last; need = (size_t)N * sizeof(double); }
    else if (!strcmp(field, "substeps")) { src = p->dev.nsub; need = (size_t)N * sizeof(uint64_t); }
    else return gw_set_error(GW_EFIELD, "unknown plant field %s", field);
    if (bytes != need) return gw_set_error(GW_EFIELD, "field %s needs %zu bytes, got %zu", field, need, bytes);
    PLANT_HIP(hipMemcpy(dst, src, need, hipMemcpyDeviceToHost), (void)0);
    return GW_OK;
}

} // extern "C"
