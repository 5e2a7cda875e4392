// runtime.hip -- device selection, streams, raw buffers (C ABI: include/cg_hip.h "runtime").
#include <cstdarg>

#include "cgh_internal.h"

cgh_context g_cgh;

int cgh_noalloc_setup(int device);  // cg.hip

void cgh_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_cgh.err, sizeof g_cgh.err, fmt, ap);
    va_end(ap);
    if (!getenv("CG_HIP_SILENT_ERRORS")) fprintf(stderr, "libcg_hip: %s\n", g_cgh.err);
}

// Values written by the cgh_config_* setters, by field offset: they outlive cgh_init and cgh_config_reload.
namespace {
struct cfg_override {
    size_t  off;
    int     bytes;
    int64_t v;
};
cfg_override g_over[32];
int          g_nover = 0;
void record_override(size_t off, int bytes, int64_t v)
{
    for (int k = 0; k < g_nover; ++k)
        if (g_over[k].off == off) {
            g_over[k].v = v;
            return;
        }
    if (g_nover < 32) g_over[g_nover++] = {off, bytes, v};
}
} // namespace
void cgh_config_set(int *field, int v)
{
    *field = v;
    record_override((size_t)((char *)field - (char *)&g_cfg), 4, v);
}
void cgh_config_set(int64_t *field, int64_t v)
{
    *field = v;
    record_override((size_t)((char *)field - (char *)&g_cfg), 8, v);
}

// The CGH_* knobs (DESIGN.md lists them): read by cgh_init and cgh_config_reload only.
static void read_env_knobs()
{
    auto env_int = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    g_cfg = cgh_config();
    g_cfg.ell_generic = env_int("CGH_ELL_GENERIC", g_cfg.ell_generic);
    g_cfg.ell_dict = env_int("CGH_ELL_DICT", g_cfg.ell_dict);
    g_cfg.ell_blocked = env_int("CGH_ELL_BLOCKED", g_cfg.ell_blocked);
    g_cfg.ell_stage = env_int("CGH_ELL_STAGE", g_cfg.ell_stage);
    g_cfg.ell_stage_tiles = env_int("CGH_ELL_STAGE_TILES", g_cfg.ell_stage_tiles);
    g_cfg.ell_variant = env_int("CGH_ELL_VARIANT", g_cfg.ell_variant);
    if (getenv("CGH_LLC_MB")) g_cfg.llc_mb = atof(getenv("CGH_LLC_MB"));
    g_cfg.csr_window = env_int("CGH_CSR_WINDOW", g_cfg.csr_window);
    if (g_cfg.csr_window != 512 && g_cfg.csr_window != 2048) g_cfg.csr_window = 1024;
    g_cfg.csr_exact = env_int("CGH_CSR_EXACT", g_cfg.csr_exact);
    if (getenv("CGH_XSUM_MIN_ROWS")) g_cfg.xsum_min_rows = atoll(getenv("CGH_XSUM_MIN_ROWS"));
    g_cfg.xsum_xupd = env_int("CGH_XSUM_XUPD", g_cfg.xsum_xupd);
    g_cfg.xsum_fused = env_int("CGH_XSUM_FUSED", g_cfg.xsum_fused);
    g_cfg.serpentine = env_int("CGH_SERPENTINE", g_cfg.serpentine);
    g_cfg.vec_variant = env_int("CGH_VEC_VARIANT", g_cfg.vec_variant);
    g_cfg.graph = env_int("CGH_GRAPH", g_cfg.graph);
    g_cfg.persistent = env_int("CGH_PERSISTENT", g_cfg.persistent);
    g_cfg.ell_wgs = env_int("CGH_ELL_WGS", g_cfg.ell_wgs);
    g_cfg.xsum_tail = env_int("CGH_XSUM_TAIL", g_cfg.xsum_tail);
    if (getenv("CGH_GRAPH_MAX_ROWS")) g_cfg.graph_max_rows = atoll(getenv("CGH_GRAPH_MAX_ROWS"));
    g_cfg.csrb_rpw = env_int("CGH_CSRB_RPW", g_cfg.csrb_rpw);
    g_cfg.csrb_slab_bits = env_int("CGH_CSRB_SLAB_BITS", g_cfg.csrb_slab_bits);
    g_cfg.csrb_wgs_per_cu = env_int("CGH_CSRB_WGS", g_cfg.csrb_wgs_per_cu);
    g_cfg.csrb_window = env_int("CGH_CSRB_WINDOW", g_cfg.csrb_window);
    g_cfg.csrb_quorum_div = env_int("CGH_CSRB_QUORUM", g_cfg.csrb_quorum_div);
    g_cfg.csrb_lds_atomics = env_int("CGH_CSRB_LDS_ATOMICS", g_cfg.csrb_lds_atomics);
    g_cfg.sell_sigma = env_int("CGH_SELL_SIGMA", g_cfg.sell_sigma);
    g_cfg.pb_span_bits = env_int("CGH_PB_SPAN_BITS", g_cfg.pb_span_bits);
    if (getenv("CGH_PB_TARGET")) g_cfg.pb_target_entries = atoll(getenv("CGH_PB_TARGET"));
    g_cfg.pb_split = env_int("CGH_PB_SPLIT", g_cfg.pb_split);
    g_cfg.pb_ordered = env_int("CGH_PB_ORDERED", g_cfg.pb_ordered);
    // what the host set through cgh_config_* wins over defaults and environment
    for (int k = 0; k < g_nover; ++k) {
        char *f = (char *)&g_cfg + g_over[k].off;
        if (g_over[k].bytes == 4) *(int *)f = (int)g_over[k].v;
        else *(int64_t *)f = g_over[k].v;
    }
}

extern "C" {

const char *cgh_last_error(void) { return g_cgh.err; }

int cgh_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

int cgh_init(int device)
{
    if (g_cgh.ready && g_cgh.device == device) return CGH_OK;
    if (g_cgh.ready) cgh_shutdown();
    // The multi-rank transports share device memory between processes (hipIpc, RCCL); hosts whose driver only supports
    // dmabuf IPC need this set BEFORE the runtime initialises (otherwise hipIpcGetMemHandle: invalid argument).  An
    // explicit setting of the caller's is left alone.
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        cgh_set_error("cgh_init: no HIP device visible (%s)", e == hipSuccess ? "count=0" : hipGetErrorString(e));
        return CGH_ENODEV;
    }
    if (device < 0 || device >= count) {
        cgh_set_error("cgh_init: device %d out of range [0,%d)", device, count);
        return CGH_EINVAL;
    }
    CGH_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    CGH_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        cgh_set_error("cgh_init: device %d is %s; this library carries gfx950 (MI355X) code only", device, prop.gcnArchName);
        return CGH_ENODEV;
    }
    g_cgh.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    read_env_knobs();  // the environment is read here, once, never in a launch path
    {
        const int rc = cgh_noalloc_setup(device);  // address of the built-in workspace on THIS device (cg.hip)
        if (rc) return rc;
    }
    CGH_HIP(hipStreamCreateWithFlags(&g_cgh.stream, hipStreamNonBlocking));
    CGH_HIP(hipStreamCreateWithFlags(&g_cgh.comm_stream, hipStreamNonBlocking));
    g_cgh.device = device;
    g_cgh.ready = true;
    return CGH_OK;
}

void cgh_shutdown(void)
{
    if (!g_cgh.ready) return;
    (void)hipDeviceSynchronize();
    if (g_cgh.stream) (void)hipStreamDestroy(g_cgh.stream);
    if (g_cgh.comm_stream) (void)hipStreamDestroy(g_cgh.comm_stream);
    g_cgh.stream = g_cgh.comm_stream = nullptr;
    g_cgh.ready = false;
    g_cgh.device = -1;
}

// Re-read the CGH_* environment knobs (cgh_init reads them once; tests that change the environment between two solves call this).
int cgh_config_reload(void)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    read_env_knobs();
    return CGH_OK;
}

// Forget what the cgh_config_* setters set and read defaults + environment again.
int cgh_config_reset(void)
{
    g_nover = 0;
    if (g_cgh.ready) read_env_knobs();
    return CGH_OK;
}

int cgh_config_sell_sigma(int sigma_rows)
{
    if (sigma_rows < 0) return CGH_EINVAL;
    if (sigma_rows) cgh_config_set(&g_cfg.sell_sigma, sigma_rows < 64 ? 64 : sigma_rows);
    return CGH_OK;
}

int cgh_config_persistent(int on)
{
    cgh_config_set(&g_cfg.persistent, on ? 1 : 0);
    return CGH_OK;
}

int cgh_sync(void)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    CGH_HIP(hipStreamSynchronize(g_cgh.stream));
    CGH_HIP(hipStreamSynchronize(g_cgh.comm_stream));
    return CGH_OK;
}

static long long g_alloc_calls = 0;
/* how often the library has called hipMalloc through cgh_alloc since it was loaded (tests: a call that must not allocate) */
long long cgh_alloc_calls(void) { return g_alloc_calls; }

int cgh_mem_info(size_t *free_bytes, size_t *total_bytes)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    size_t f = 0, t = 0;
    CGH_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return CGH_OK;
}

/* Capture what the library queues on its compute stream from here on (global mode: an allocation or a synchronisation
 * inside the captured region is an ERROR of the runtime, not a slow path) ... */
int cgh_capture_begin(void)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    CGH_HIP(hipStreamBeginCapture(g_cgh.stream, hipStreamCaptureModeGlobal));
    return CGH_OK;
}
/* ... and end the capture, run the captured graph `replays` times on the same stream, and release it. */
int cgh_capture_end_launch(int replays)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    hipGraph_t     g = nullptr;
    hipGraphExec_t ex = nullptr;
    CGH_HIP(hipStreamEndCapture(g_cgh.stream, &g));
    if (!g) {
        cgh_set_error("cgh_capture_end_launch: the capture was invalidated");
        return CGH_EHIP;
    }
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    for (int k = 0; e == hipSuccess && k < replays; ++k) e = hipGraphLaunch(ex, g_cgh.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_cgh.stream);
    if (ex) (void)hipGraphExecDestroy(ex);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        cgh_set_error("cgh_capture_end_launch: %s", hipGetErrorString(e));
        return CGH_EHIP;
    }
    return CGH_OK;
}

void *cgh_alloc(size_t bytes)
{
    CGH_REQUIRE_READY(nullptr);
    void *p = nullptr;
    if (bytes == 0) bytes = 16;
    ++g_alloc_calls;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        cgh_set_error("cgh_alloc(%zu): %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

void cgh_free(void *d_ptr)
{
    if (d_ptr && g_cgh.ready) (void)hipFree(d_ptr);
}

int cgh_h2d(void *d_dst, const void *h_src, size_t bytes)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (bytes == 0) return CGH_OK;
    CGH_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, g_cgh.stream));
    CGH_HIP(hipStreamSynchronize(g_cgh.stream));
    return CGH_OK;
}

int cgh_d2h(void *h_dst, const void *d_src, size_t bytes)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (bytes == 0) return CGH_OK;
    CGH_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, g_cgh.stream));
    CGH_HIP(hipStreamSynchronize(g_cgh.stream));
    return CGH_OK;
}

int cgh_d2d(void *d_dst, const void *d_src, size_t bytes)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (bytes == 0) return CGH_OK;
    CGH_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, g_cgh.stream));
    return CGH_OK;
}

int cgh_memset(void *d_dst, int byte, size_t bytes)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (bytes == 0) return CGH_OK;
    CGH_HIP(hipMemsetAsync(d_dst, byte, bytes, g_cgh.stream));
    return CGH_OK;
}

} // extern "C"
