// gw_ctrl_api.cpp -- C-ABI of the closed control loop (include/gymwipe_amd.h, "Control loop"): host side.
#include "gw_internal.h"

#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <vector>

int gw_set_error(int code, const char* fmt, ...);      // gw_api.cpp

struct gw_ctrl {
    gw_ctrl_config cfg;
    GwHostTables tab;
    GwCtrlDev dev;
    void* blocks[24];
    int nblocks;
};

namespace {

#define CTRL_HIP(expr, cleanup)                                                                   \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            cleanup;                                                                              \
            return gw_set_error(GW_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e));          \
        }                                                                                         \
    } while (0)

template <class T>
int calloc_dev(gw_ctrl* c, T** out, size_t count)
{
    void* q = nullptr;
    if (c->nblocks >= (int)(sizeof c->blocks / sizeof c->blocks[0]) || hipMalloc(&q, count * sizeof(T)) != hipSuccess)
        return gw_set_error(GW_ENOMEM, "hipMalloc failed in gw_ctrl_create");
    c->blocks[c->nblocks++] = q;
    *out = (T*)q;
    return GW_OK;
}

} // namespace

extern "C" {

int gw_ctrl_config_default(gw_ctrl_config* c, int64_t num_envs)
{
    if (!c) return gw_set_error(GW_EINVAL, "cfg is NULL");
    memset(c, 0, sizeof *c);
    int rc = gw_config_default(&c->net, num_envs, 3);
    if (rc) return rc;
    // envs/inverted_pendulum.py:75-93: controller at (0, -1), RRM at (0, 1), sensor and actuator on the wagon.  The
    // actuator sits half a metre along the wagon here: two co-located radios would keep the reference's "attenuation 0".
    c->net.pos[0][0] = 0.0; c->net.pos[0][1] = 0.0;
    c->net.pos[1][0] = 0.0; c->net.pos[1][1] = -1.0;
    c->net.pos[2][0] = 0.5; c->net.pos[2][1] = 0.0;
    c->net.pos[3][0] = 0.0; c->net.pos[3][1] = 1.0;
    c->net.mult[0] = 1; c->net.mult[1] = 0; c->net.mult[2] = 0;
    c->net.dest[0] = 1; c->net.dest[1] = 2; c->net.dest[2] = 0;
    gw_plant_config pc;
    rc = gw_plant_config_default(&pc, num_envs);
    if (rc) return rc;
    memcpy(c->A, pc.A, sizeof c->A);
    for (int i = 0; i < 4; ++i) { c->B[i] = -pc.B[i]; c->x0[i] = pc.x0[i]; }   // input sign: the reference's control law (u = -angle) damps
    c->u0 = pc.u0;
    c->ctrl_start_tick = 20;
    c->ctrl_period_ticks = 10;                  // 10 ms, control/inverted_pendulum.py:69
    return GW_OK;
}

int gw_ctrl_destroy(gw_ctrl* c)
{
    if (!c) return GW_OK;
    (void)hipSetDevice(c->cfg.net.hip_device);
    for (int i = 0; i < c->nblocks; ++i) (void)hipFree(c->blocks[i]);
    delete c;
    return GW_OK;
}

int gw_ctrl_create(const gw_ctrl_config* cfg, gw_ctrl** out)
{
    if (!cfg || !out) return gw_set_error(GW_EINVAL, "cfg/out is NULL");
    *out = nullptr;
    if (cfg->net.num_devices != 3) return gw_set_error(GW_EINVAL, "the control loop has three network devices (sensor, controller, actuator)");
    if (cfg->ctrl_period_ticks < 1 || cfg->ctrl_start_tick < 0) return gw_set_error(GW_EINVAL, "ctrl_start_tick / ctrl_period_ticks out of range");
    int rc = gw_validate_config(cfg->net);
    if (rc) return rc;
    {
        // the step kernel stages a step's queue appends in LDS, 24 per queue: a step must not contain more counter ticks
        const double step_max = ((double)(cfg->net.max_duration - 1) * cfg->net.duration_factor + 3.0) * cfg->net.slot
                                + (double)(cfg->net.mac_header_bytes + 16) * 8.0 / (cfg->net.code_rate * cfg->net.bit_rate);
        if (step_max / cfg->net.counter_interval + 2.0 > 24.0)
            return gw_set_error(GW_EUNSUPPORTED, "more than 24 counter ticks can fall into one step with this duration range / counter interval");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return gw_set_error(GW_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
    if (cfg->net.hip_device < 0 || cfg->net.hip_device >= ndev) return gw_set_error(GW_EINVAL, "hip_device out of range");
    gw_ctrl* c = new (std::nothrow) gw_ctrl();
    if (!c) return gw_set_error(GW_ENOMEM, "out of host memory");
    memset(c, 0, sizeof *c);
    c->cfg = *cfg;
    char msg[256] = "";
    rc = gw_build_tables(c->cfg.net, c->tab, msg, sizeof msg);
    if (!rc && c->tab.overflow) rc = GW_EUNSUPPORTED;        // (the control-loop kernel keeps the byte noise states)
    if (rc) { delete c; return gw_set_error(rc, "%s", msg); }
    CTRL_HIP(hipSetDevice(cfg->net.hip_device), delete c);
    GwDevConst k;
    memset(&k, 0, sizeof k);
    gw_fill_dev_const(c->cfg.net, c->tab, k);
    const int64_t N = cfg->net.num_envs;
    const int R = 4;
    GwCtrlDev& d = c->dev;
    d.N = N;
    memcpy(d.A, cfg->A, sizeof d.A);
    memcpy(d.B, cfg->B, sizeof d.B);
    d.start = (uint32_t)cfg->ctrl_start_tick;
    d.period = (uint32_t)cfg->ctrl_period_ticks;
    GwDevConst* d_cst = nullptr; uint8_t* d_trans = nullptr; double* d_ber = nullptr;
    const size_t tcount = (size_t)R * R * GW_MAX_NSTATES;
#define CA(ptr, cnt) do { rc = calloc_dev(c, &(ptr), (size_t)(cnt)); if (rc) { gw_ctrl_destroy(c); return rc; } } while (0)
    CA(d.now, N); CA(d.wake, N); CA(d.x, N * 4); CA(d.u, N); CA(d.ang, N);
    CA(d.ktick, N); CA(d.got, N * 2); CA(d.ntx, N); CA(d.ncmd, N); CA(d.nsub, N); CA(d.flags, N);
    CA(d.qhl, N * 2); CA(d.rxs, N * R); CA(d.pay, N * 2 * GW_RING_PHYS);
    CA(d_cst, 1); CA(d_trans, tcount); CA(d_ber, tcount);
#undef CA
    d.cst = d_cst; d.trans = d_trans; d.ber = d_ber;
    CTRL_HIP(hipMemcpy(d_cst, &k, sizeof k, hipMemcpyHostToDevice), gw_ctrl_destroy(c));
    CTRL_HIP(hipMemcpy(d_trans, c->tab.trans, tcount, hipMemcpyHostToDevice), gw_ctrl_destroy(c));
    CTRL_HIP(hipMemcpy(d_ber, c->tab.ber, tcount * sizeof(double), hipMemcpyHostToDevice), gw_ctrl_destroy(c));
    CTRL_HIP(hipMemset(d.pay, 0, (size_t)N * 2 * GW_RING_PHYS * sizeof(double)), gw_ctrl_destroy(c));
    if (gw_ctrl_launch_init(d, cfg->x0, cfg->u0, nullptr)) { gw_ctrl_destroy(c); return gw_set_error(GW_EHIP, "control-loop init launch failed"); }
    CTRL_HIP(hipDeviceSynchronize(), gw_ctrl_destroy(c));
    *out = c;
    return GW_OK;
}

int gw_ctrl_step(gw_ctrl* c, const int32_t* device_dev, const int32_t* duration_dev, int32_t* obs_dev, float* reward_dev,
                 double* angle_deg_dev, void* stream)
{
    if (!c) return gw_set_error(GW_EINVAL, "handle is NULL");
    if (!device_dev || !duration_dev || !obs_dev || !reward_dev) return gw_set_error(GW_EINVAL, "gw_ctrl_step: NULL device pointer");
    CTRL_HIP(hipSetDevice(c->cfg.net.hip_device), (void)0);
    if (gw_ctrl_launch_step(c->dev, device_dev, duration_dev, obs_dev, reward_dev, angle_deg_dev, stream))
        return gw_set_error(GW_EHIP, "control-loop step launch failed");
    return GW_OK;
}

int gw_ctrl_get_state(gw_ctrl* c, const char* field, void* dst, size_t bytes)
{
    if (!c || !field || !dst) return gw_set_error(GW_EINVAL, "handle/field/dst is NULL");
    CTRL_HIP(hipSetDevice(c->cfg.net.hip_device), (void)0);
    CTRL_HIP(hipDeviceSynchronize(), (void)0);
    const GwCtrlDev& d = c->dev;
    const int64_t N = d.N;
    const void* src = nullptr; size_t need = 0;
    if (!strcmp(field, "now")) { src = d.now; need = N * sizeof(double); }
    else if (!strcmp(field, "wake")) { src = d.wake; need = N * sizeof(double); }
    else if (!strcmp(field, "x")) { src = d.x; need = N * 4 * sizeof(double); }
    else if (!strcmp(field, "u")) { src = d.u; need = N * sizeof(double); }
    else if (!strcmp(field, "angle_deg")) { src = d.ang; need = N * sizeof(double); }
    else if (!strcmp(field, "received")) { src = d.got; need = N * 2 * sizeof(uint32_t); }
    else if (!strcmp(field, "n_tx")) { src = d.ntx; need = N * sizeof(uint32_t); }
    else if (!strcmp(field, "commands")) { src = d.ncmd; need = N * sizeof(uint32_t); }
    else if (!strcmp(field, "substeps")) { src = d.nsub; need = N * sizeof(uint32_t); }
    else if (!strcmp(field, "flags")) { src = d.flags; need = N * sizeof(uint32_t); }
    if (src) {
        if (bytes != need) return gw_set_error(GW_EFIELD, "field %s needs %zu bytes, got %zu", field, need, bytes);
        CTRL_HIP(hipMemcpy(dst, src, need, hipMemcpyDeviceToHost), (void)0);
        return GW_OK;
    }
    if (!strcmp(field, "qlen")) {                          // int32[N][2]: sensor, controller
        if (bytes != (size_t)N * 2 * sizeof(int32_t)) return gw_set_error(GW_EFIELD, "field qlen needs %zu bytes", (size_t)N * 2 * sizeof(int32_t));
        std::vector<uint16_t> hl((size_t)N * 2);
        CTRL_HIP(hipMemcpy(hl.data(), d.qhl, hl.size() * sizeof(uint16_t), hipMemcpyDeviceToHost), (void)0);
        for (int64_t e = 0; e < N; ++e) for (int q = 0; q < 2; ++q) ((int32_t*)dst)[e * 2 + q] = hl[(size_t)q * N + e] >> 8;
        return GW_OK;
    }
    if (!strcmp(field, "rx_power")) {                      // f64[N][4]
        if (bytes != (size_t)N * 4 * sizeof(double)) return gw_set_error(GW_EFIELD, "field rx_power needs %zu bytes", (size_t)N * 4 * sizeof(double));
        std::vector<uint8_t> s((size_t)N * 4);
        CTRL_HIP(hipMemcpy(s.data(), d.rxs, s.size(), hipMemcpyDeviceToHost), (void)0);
        for (int64_t e = 0; e < N; ++e) for (int r = 0; r < 4; ++r) ((double*)dst)[e * 4 + r] = c->tab.state_val[r][s[(size_t)r * N + e]];
        return GW_OK;
    }
    return gw_set_error(GW_EFIELD, "unknown control-loop field %s", field);
}

} // extern "C"
