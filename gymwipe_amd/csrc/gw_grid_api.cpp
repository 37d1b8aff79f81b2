// gw_grid_api.cpp -- C-ABI of the PHY grid (include/gymwipe_amd.h, "PHY grid"): host side.
#include "gw_internal.h"

#include <hip/hip_runtime.h>

#include <math.h>
#include <string.h>
#include <new>
#include <vector>

int gw_set_error(int code, const char* fmt, ...);      // gw_api.cpp

struct gw_grid {
    gw_grid_config cfg;
    GwGridDev dev;
    void* blocks[6];
    int nblocks;
};

#define GRID_HIP(expr, cleanup)                                                                   \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            cleanup;                                                                              \
            return gw_set_error(GW_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e));          \
        }                                                                                         \
    } while (0)

extern "C" {

int gw_grid_config_default(gw_grid_config* c, int64_t num_envs, int32_t n)
{
    if (!c) return gw_set_error(GW_EINVAL, "cfg is NULL");
    if (n < 1 || n > GW_GRID_MAX_DEVICES) return gw_set_error(GW_EINVAL, "num_devices must be in [1, %d]", GW_GRID_MAX_DEVICES);
    memset(c, 0, sizeof *c);
    c->abi_version = GW_ABI_VERSION;
    c->num_envs = num_envs;
    c->num_devices = n;
    const int cols = (int)sqrt((double)n);                       // tests/test_benchmark.py:64
    for (int i = 0; i < n; ++i) { c->pos[i][0] = (double)i / cols; c->pos[i][1] = (double)(i % cols); }   // :68
    c->slot = 1e-6; c->frequency = 2.4e9; c->bandwidth = 22e6; c->temperature_c = 20.0;
    c->bit_rate = 133.33333e3; c->code_rate = 0.75; c->max_ber = 0.25;
    c->tx_power_dbm = 40.0;                                      // :47
    c->send_interval = 1e-2;                                     // :17
    c->header_bytes = 13;                                        // SimpleMacHeader
    c->payload_bytes = 26;                                       // "A message to all my homies" (:44)
    c->move_interval = 1e-3;                                     // :18
    c->move_span = 0.2;                                          // :79-80
    c->seed = 0;
    return GW_OK;
}

int gw_grid_create(const gw_grid_config* cfg, const double* delays, gw_grid** out)
{
    if (!cfg || !delays || !out) return gw_set_error(GW_EINVAL, "cfg/delays/out is NULL");
    *out = nullptr;
    if (cfg->abi_version != GW_ABI_VERSION) return gw_set_error(GW_EINVAL, "abi_version mismatch");
    const int n = cfg->num_devices;
    const int64_t N = cfg->num_envs;
    if (n < 1 || n > GW_GRID_MAX_DEVICES || N <= 0) return gw_set_error(GW_EINVAL, "num_devices / num_envs out of range");
    if (cfg->max_ber != 0.25 || (2 - cfg->code_rate) * 8 != floor((2 - cfg->code_rate) * 8))
        return gw_set_error(GW_EUNSUPPORTED, "the grid kernel implements the 3/4-rate decision rule only");
    for (int64_t i = 0; i < N * n; ++i)
        if (!(delays[i] >= 0)) return gw_set_error(GW_EINVAL, "initial delays must be non-negative");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return gw_set_error(GW_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
    if (cfg->hip_device < 0 || cfg->hip_device >= ndev) return gw_set_error(GW_EINVAL, "hip_device out of range");

    gw_grid* g = new (std::nothrow) gw_grid();
    if (!g) return gw_set_error(GW_ENOMEM, "out of host memory");
    memset(g, 0, sizeof *g);
    g->cfg = *cfg;
    GRID_HIP(hipSetDevice(cfg->hip_device), delete g);

    // static link table through the same libm calls as the reference (cf. gw_tables.cpp)
    std::vector<double> prx((size_t)n * n, 0.0);
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            if (a == b) continue;
            const double ax = cfg->pos[a][0], ay = cfg->pos[a][1], bx = cfg->pos[b][0], by = cfg->pos[b][1];
            double att = 0.0;
            if (!(ax == bx && ay == by))
                att = 20 * log10(sqrt(pow(ax - bx, 2.0) + pow(ay - by, 2.0))) + 20 * log10(cfg->frequency) - 147.55;
            prx[(size_t)a * n + b] = pow(10.0, (cfg->tx_power_dbm - att) / 10);
        }
    const double thermal = 1.38e-23 * (cfg->temperature_c + 273.15) * cfg->bandwidth * 1000;
    const double data_rate = cfg->code_rate * cfg->bit_rate, cf = 2 - cfg->code_rate;

    GwGridDev& d = g->dev;
    d.N = N; d.n = n;
    double* d_prx = nullptr; double* d_delays = nullptr; double* d_pos = nullptr; double* d_txp = nullptr;
    auto alloc = [&](void** p, size_t bytes) {
        if (hipMalloc(p, bytes) != hipSuccess) return false;
        g->blocks[g->nblocks++] = *p;
        return true;
    };
    if (!alloc((void**)&d.lanes, (size_t)N * n * sizeof(GwGridLane)) || !alloc((void**)&d.envs, (size_t)N * sizeof(GwGridEnv)) ||
        !alloc((void**)&d_prx, prx.size() * sizeof(double))) {
        gw_grid_destroy(g);
        return gw_set_error(GW_ENOMEM, "hipMalloc failed in gw_grid_create");
    }
    if (cfg->mobile && !alloc((void**)&d_txp, (size_t)N * n * n * sizeof(double))) { gw_grid_destroy(g); return gw_set_error(GW_ENOMEM, "hipMalloc failed"); }
    if (hipMalloc((void**)&d_delays, ((size_t)N * n + 2 * (size_t)n) * sizeof(double)) != hipSuccess) { gw_grid_destroy(g); return gw_set_error(GW_ENOMEM, "hipMalloc failed"); }
    d_pos = d_delays + (size_t)N * n;
    d.prx = d_prx;
    d.slot = cfg->slot; d.send_interval = cfg->send_interval; d.bit_rate = cfg->bit_rate;
    {
        volatile double hb = (double)(cfg->header_bytes * 8), pb = (double)(cfg->payload_bytes * 8);
        d.hdr_bits = hb * cf; d.pay_bits = pb * cf;          // physical.py:259-263
        d.hdr_dur = hb / data_rate; d.pay_dur = pb / data_rate;   // :244-247
    }
    d.max_events = 0;                                   // set per run from the simulated span
    d.mobile = cfg->mobile ? 1u : 0u; d.move_interval = cfg->move_interval; d.move_span = cfg->move_span;
    d.tx_power_dbm = cfg->tx_power_dbm; d.twenty_log_f = 20 * log10(cfg->frequency); d.txp = d_txp; d.seed = cfg->seed;
    d.ten_log_br = 10 * log10(cfg->bit_rate);
    d.sqrt2pi = sqrt(2 * M_PI);
    GRID_HIP(hipMemcpy(d_prx, prx.data(), prx.size() * sizeof(double), hipMemcpyHostToDevice), ((void)hipFree(d_delays), gw_grid_destroy(g)));
    GRID_HIP(hipMemcpy(d_delays, delays, (size_t)N * n * sizeof(double), hipMemcpyHostToDevice), ((void)hipFree(d_delays), gw_grid_destroy(g)));
    {
        std::vector<double> pos((size_t)2 * n);
        for (int i = 0; i < n; ++i) { pos[2 * i] = cfg->pos[i][0]; pos[2 * i + 1] = cfg->pos[i][1]; }
        GRID_HIP(hipMemcpy(d_pos, pos.data(), pos.size() * sizeof(double), hipMemcpyHostToDevice), ((void)hipFree(d_delays), gw_grid_destroy(g)));
        if (d_txp) GRID_HIP(hipMemset(d_txp, 0, (size_t)N * n * n * sizeof(double)), ((void)hipFree(d_delays), gw_grid_destroy(g)));
    }
    if (gw_grid_launch_init(d, d_delays, d_pos, thermal, nullptr)) { (void)hipFree(d_delays); gw_grid_destroy(g); return gw_set_error(GW_EHIP, "grid init launch failed"); }
    GRID_HIP(hipDeviceSynchronize(), ((void)hipFree(d_delays), gw_grid_destroy(g)));
    (void)hipFree(d_delays);
    *out = g;
    return GW_OK;
}

int gw_grid_destroy(gw_grid* g)
{
    if (!g) return GW_OK;
    (void)hipSetDevice(g->cfg.hip_device);
    for (int i = 0; i < g->nblocks; ++i) (void)hipFree(g->blocks[i]);
    delete g;
    return GW_OK;
}

int gw_grid_run(gw_grid* g, double seconds, void* stream)
{
    if (!g) return gw_set_error(GW_EINVAL, "grid is NULL");
    if (!(seconds > 0)) return gw_set_error(GW_EINVAL, "seconds must be positive");
    GRID_HIP(hipSetDevice(g->cfg.hip_device), (void)0);
    // ~12 events per packet, one packet per device and send interval; generous factor on top
    const double expect = 12.0 * g->dev.n * (seconds / g->cfg.send_interval + 2.0)
                        + (g->cfg.mobile ? g->dev.n * (seconds / g->cfg.move_interval + 2.0) : 0.0);
    g->dev.max_events = (uint32_t)(expect * 8.0 < 4.0e9 ? expect * 8.0 + 1000.0 : 4.0e9);
    if (gw_grid_launch_run(g->dev, seconds, stream)) return gw_set_error(GW_EHIP, "grid run launch failed");
    return GW_OK;
}

int gw_grid_set_position(gw_grid* g, int32_t device, const double* x_host, const double* y_host, void* stream)
{
    if (!g || !x_host || !y_host) return gw_set_error(GW_EINVAL, "grid/x/y is NULL");
    if (!g->cfg.mobile) return gw_set_error(GW_EUNSUPPORTED, "gw_grid_set_position needs cfg.mobile (per-replica geometry)");
    if (device < 0 || device >= g->dev.n) return gw_set_error(GW_EINVAL, "device out of range");
    GRID_HIP(hipSetDevice(g->cfg.hip_device), (void)0);
    const int64_t N = g->dev.N;
    double* d_xy = nullptr;
    GRID_HIP(hipMalloc((void**)&d_xy, (size_t)2 * N * sizeof(double)), (void)0);
    GRID_HIP(hipMemcpy(d_xy, x_host, (size_t)N * sizeof(double), hipMemcpyHostToDevice), (void)hipFree(d_xy));
    GRID_HIP(hipMemcpy(d_xy + N, y_host, (size_t)N * sizeof(double), hipMemcpyHostToDevice), (void)hipFree(d_xy));
    const int rc = gw_grid_launch_set_position(g->dev, device, d_xy, d_xy + N, stream);
    GRID_HIP(hipStreamSynchronize((hipStream_t)stream), (void)hipFree(d_xy));
    (void)hipFree(d_xy);
    if (rc) return gw_set_error(GW_EHIP, "grid set_position launch failed");
    return GW_OK;
}

int gw_grid_get_state(gw_grid* g, const char* field, void* dst, size_t bytes)
{
    if (!g || !field || !dst) return gw_set_error(GW_EINVAL, "grid/field/dst is NULL");
    GRID_HIP(hipSetDevice(g->cfg.hip_device), (void)0);
    GRID_HIP(hipDeviceSynchronize(), (void)0);
    const int64_t N = g->dev.N; const int n = g->dev.n;
    std::vector<GwGridLane> lanes((size_t)N * n);
    std::vector<GwGridEnv> envs((size_t)N);
    GRID_HIP(hipMemcpy(lanes.data(), g->dev.lanes, lanes.size() * sizeof(GwGridLane), hipMemcpyDeviceToHost), (void)0);
    GRID_HIP(hipMemcpy(envs.data(), g->dev.envs, envs.size() * sizeof(GwGridEnv), hipMemcpyDeviceToHost), (void)0);
#define NEEDB(cnt, type) if (bytes != (size_t)(cnt) * sizeof(type)) return gw_set_error(GW_EFIELD, "field %s: size mismatch", field)
    if (!strcmp(field, "now")) { NEEDB(N, double); for (int64_t e = 0; e < N; ++e) ((double*)dst)[e] = envs[e].now; return GW_OK; }
    if (!strcmp(field, "events")) { NEEDB(N, uint32_t); for (int64_t e = 0; e < N; ++e) ((uint32_t*)dst)[e] = envs[e].events; return GW_OK; }
    if (!strcmp(field, "n_tx")) { NEEDB(N, uint32_t); for (int64_t e = 0; e < N; ++e) ((uint32_t*)dst)[e] = envs[e].n_tx; return GW_OK; }
    if (!strcmp(field, "on_air")) {                     // len(frequencyBand.getActiveTransmissions())
        NEEDB(N, uint32_t);
        for (int64_t e = 0; e < N; ++e) { uint32_t c = 0; for (int i = 0; i < n; ++i) c += lanes[(size_t)e * n + i].tx_on ? 1u : 0u; ((uint32_t*)dst)[e] = c; }
        return GW_OK;
    }
    if (!strcmp(field, "flags")) {
        NEEDB(N, uint32_t);
        for (int64_t e = 0; e < N; ++e) { uint32_t f = 0; for (int i = 0; i < n; ++i) f |= lanes[(size_t)e * n + i].flags; ((uint32_t*)dst)[e] = f; }
        return GW_OK;
    }
    if (!strcmp(field, "pos")) {
        NEEDB(N * n * 2, double);
        for (size_t i = 0; i < lanes.size(); ++i) { ((double*)dst)[2 * i] = lanes[i].px; ((double*)dst)[2 * i + 1] = lanes[i].py; }
        return GW_OK;
    }
    if (!strcmp(field, "rx_power")) { NEEDB(N * n, double); for (size_t i = 0; i < lanes.size(); ++i) ((double*)dst)[i] = lanes[i].rx_power; return GW_OK; }
    const char* names[5] = {"n_sent", "hdr_ok", "hdr_fail", "pay_ok", "pay_fail"};
    for (int k = 0; k < 5; ++k)
        if (!strcmp(field, names[k])) {
            NEEDB(N * n, uint32_t);
            for (size_t i = 0; i < lanes.size(); ++i) {
                const GwGridLane& L = lanes[i];
                ((uint32_t*)dst)[i] = k == 0 ? L.n_sent : k == 1 ? L.hdr_ok : k == 2 ? L.hdr_fail : k == 3 ? L.pay_ok : L.pay_fail;
            }
            return GW_OK;
        }
#undef NEEDB
    return gw_set_error(GW_EFIELD, "unknown grid field %s", field);
}

} // extern "C"
