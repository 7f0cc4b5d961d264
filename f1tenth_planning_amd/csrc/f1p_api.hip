// f1p_api.hip -- the extern "C" surface of libf1p.so (include/f1p.h): context, device memory, scene upload,
// host-pointer wrappers around the kernel launchers, and the RCCL exchange step.
#include <dlfcn.h>
#include <math.h>
#include <string.h>

#include <algorithm>

#include <new>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "f1p_internal.h"

static thread_local std::string g_create_error;

namespace f1p {

int set_error(f1p_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

int check_hip(f1p_ctx* ctx, hipError_t e, const char* what) {
    if (e == hipSuccess) return F1P_OK;
    const int code = (e == hipErrorOutOfMemory) ? F1P_ENOMEM : F1P_EHIP;
    return set_error(ctx, code, std::string(what) + ": " + hipGetErrorString(e));
}

int arena_reset(f1p_ctx* ctx, size_t need_bytes) {
    ctx->arena_used = 0;
    if (need_bytes <= ctx->arena_bytes) return F1P_OK;
    if (ctx->d_arena) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->d_arena); ctx->d_arena = nullptr; ctx->arena_bytes = 0; }
    size_t want = need_bytes + (need_bytes >> 2) + 4096;
    F1P_HIP(ctx, hipMalloc((void**)&ctx->d_arena, want));
    ctx->arena_bytes = want;
    return F1P_OK;
}

static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

void* arena_take(f1p_ctx* ctx, size_t bytes) {
    void* p = ctx->d_arena + ctx->arena_used;
    ctx->arena_used += al256(bytes);
    return p;
}

GridDev grid_dev(const f1p_ctx* ctx) {
    GridDev g;
    g.bits = ctx->d_bits; g.w = ctx->gw; g.h = ctx->gh; g.wwords = ctx->gwwords;
    g.inv_res = ctx->inv_res; g.ox = ctx->ox; g.oy = ctx->oy;
    return g;
}

#define F1P_SMALL_D2H_BYTES ((size_t)64 * 1024)      // calls up to this size: outputs written by the kernels straight into the page-locked block
#define F1P_BOUNCE_BYTES ((size_t)1024 * 1024)       // the block; results up to this size come back in ONE copy + a host-side scatter
static int ensure_bounce(f1p_ctx* ctx) {
    if (ctx->h_bounce) return F1P_OK;
    F1P_HIP(ctx, hipHostMalloc((void**)&ctx->h_bounce, F1P_BOUNCE_BYTES, hipHostMallocDefault));
    return F1P_OK;
}

// staged host<->device transfer plan for the *_batch wrappers
struct Stage {
    f1p_ctx* ctx;
    struct Out { void* host; void* dev; size_t bytes; };
    std::vector<Out> outs;
    size_t total = 0, out_total = 0;
    // Small calls (a single vehicle, a few hundred egos): the outputs -- and inputs of a few hundred bytes, i.e. a pose or two --
    // live in the context's page-locked, device-visible block: the kernels write their results straight into host memory and no
    // hipMemcpy is issued at all (each one is ~10-15 us of latency; a single-vehicle plan() is otherwise mostly copies).
    // Larger inputs still go through a copy: a kernel that re-reads them (previous headings, controls) must find them in HBM.
    bool zc = false;
    size_t zc_used = 0;
    explicit Stage(f1p_ctx* c) : ctx(c) {}
    void need(size_t bytes, bool used = true) { if (used) total += al256(bytes); }
    int begin() {
        zc = total <= F1P_SMALL_D2H_BYTES && ensure_bounce(ctx) == F1P_OK;
        zc_used = 0;
        return arena_reset(ctx, total);
    }
    void* zc_take(size_t bytes) { void* p = ctx->h_bounce + zc_used; zc_used += al256(bytes); return p; }
    template <typename T> int in(const T* host, size_t count, const T** dev) {
        *dev = nullptr;
        if (!host || count == 0) return F1P_OK;
        if (zc && count * sizeof(T) <= 512) {
            T* d = (T*)zc_take(count * sizeof(T));
            memcpy(d, host, count * sizeof(T));
            *dev = d;
            return F1P_OK;
        }
        T* d = (T*)arena_take(ctx, count * sizeof(T));
        F1P_HIP(ctx, hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        *dev = d;
        return F1P_OK;
    }
    template <typename T> T* out(T* host, size_t count) {
        if (!host || count == 0) return nullptr;
        T* d = zc ? (T*)zc_take(count * sizeof(T)) : (T*)arena_take(ctx, count * sizeof(T));
        outs.push_back({(void*)host, (void*)d, count * sizeof(T)});
        return d;
    }
    int finish() {
        if (zc) {
            auto in_block = [&](const void* p) { return (const char*)p >= ctx->h_bounce && (const char*)p < ctx->h_bounce + F1P_BOUNCE_BYTES; };
            for (auto& o : outs)                                  // an in/out buffer that went through the device arena (larger than a pose)
                if (!in_block(o.dev)) F1P_HIP(ctx, hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, ctx->stream));
            F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (auto& o : outs) if (in_block(o.dev)) memcpy(o.host, o.dev, o.bytes);
            return F1P_OK;
        }
        return gather(nullptr);
    }
    // Results to the host: arrays that sit back to back in the arena and together fit the page-locked block come back in ONE copy
    // and are scattered on the host (a hipMemcpyAsync per array is ~10 us of submission each -- six of them were a third of a
    // 4096-ego plan()'s latency); anything else (the trajectories) is copied on its own.  `skip`: an output the caller copies itself.
    int gather(const void* skip) {
        std::vector<Out*> small;
        const char *lo = nullptr, *hi = nullptr;
        for (auto& o : outs) {
            if (o.host == skip) continue;
            const char* b = (const char*)o.dev;
            const char* nlo = lo && lo < b ? lo : b;
            const char* nhi = hi && hi > b + o.bytes ? hi : b + o.bytes;
            if ((size_t)(nhi - nlo) <= F1P_BOUNCE_BYTES && o.bytes <= F1P_BOUNCE_BYTES / 4) { small.push_back(&o); lo = nlo; hi = nhi; }
            else F1P_HIP(ctx, hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (small.size() > 1 && ensure_bounce(ctx) == F1P_OK) {
            F1P_HIP(ctx, hipMemcpyAsync(ctx->h_bounce, lo, (size_t)(hi - lo), hipMemcpyDeviceToHost, ctx->stream));
            F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (Out* o : small) memcpy(o->host, ctx->h_bounce + ((const char*)o->dev - lo), o->bytes);
            return F1P_OK;
        }
        for (Out* o : small) F1P_HIP(ctx, hipMemcpyAsync(o->host, o->dev, o->bytes, hipMemcpyDeviceToHost, ctx->stream));
        F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return F1P_OK;
    }
};

#ifndef F1P_TRAJ_ZEROCOPY
#define F1P_TRAJ_ZEROCOPY 1
#endif
#ifndef F1P_IO_ZEROCOPY
#define F1P_IO_ZEROCOPY 1       // f1p_lattice_plan_batch: page-locked poses / result columns are read / written by the kernels themselves (0: copies, A/B)
#endif
#define F1P_PLAN_CHUNK_MIN_EGOS 2048
#ifndef F1P_ZEROCOPY_MIN_EGOS
#define F1P_ZEROCOPY_MIN_EGOS 32       // batches from this size hand page-locked caller arrays to the kernels (measured p50 plan(), fp64 rows, copies -> in place: 64 egos 0.075 -> 0.061 ms, 256: 0.090 -> 0.068, 1024: 0.126 -> 0.103, 4096: 0.236 -> 0.218; below 32 the small-call bounce block is as fast)
#endif
#define F1P_PLAN_CHUNKS_MAX 8          // = number of slice events in f1p_ctx

// The device-side address of [p, p + bytes) when the WHOLE range lies in page-locked host memory this device can address (hipHostMalloc /
// hipHostRegister), else null.  A kernel that stores through it writes straight into the caller's array; a range that is only partly
// registered would fault, so both ends must be known to HIP as host memory with ONE linear device mapping between them (the
// first byte alone had been checked, and the host address used instead of attr.devicePointer).
static void* pinned_device_ptr(const void* p, size_t bytes);
// ... with the context's own page-locked blocks answered from its table (no runtime call)
static void* pinned_device_ptr(const f1p_ctx* ctx, const void* p, size_t bytes) {
    if (!p || bytes == 0) return nullptr;
    const char* c = (const char*)p;
    if (ctx->h_step && ctx->h_step_dev && c >= ctx->h_step && c + bytes <= ctx->h_step + ctx->step_host_bytes) return ctx->h_step_dev + (c - ctx->h_step);
    for (const auto& b : ctx->host_blocks)
        if (c >= b.base && c + bytes <= b.base + b.bytes) return b.dev + (c - b.base);
    return pinned_device_ptr(p, bytes);
}
static void* pinned_device_ptr(const void* p, size_t bytes) {
    if (!p || bytes == 0) return nullptr;
    hipPointerAttribute_t a, b;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }   // pageable: not known to HIP
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
    if (bytes > 1) {
        if (hipPointerGetAttributes(&b, (const char*)p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (b.type != hipMemoryTypeHost || !b.devicePointer) return nullptr;
        if ((const char*)b.devicePointer - (const char*)a.devicePointer != (ptrdiff_t)(bytes - 1)) return nullptr;
        void* base = nullptr; size_t size = 0;                       // one allocation (where the runtime can tell)
        if (hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)a.devicePointer) == hipSuccess) {
            if ((const char*)a.devicePointer + bytes > (const char*)base + size) return nullptr;
        } else (void)hipGetLastError();
    }
    return a.devicePointer;
}

static int ensure_copy_stream(f1p_ctx* ctx) {
    if (ctx->copy_stream) return F1P_OK;
    F1P_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (auto& ev : ctx->ev_chunk) F1P_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return F1P_OK;
}

static int validate_lattice(f1p_ctx* ctx, const f1p_lattice_cfg* cfg, int E, bool device_goals, bool need_outputs_ok) {
    if (!cfg) return set_error(ctx, F1P_EINVAL, "cfg is NULL");
    if (E < 0) return set_error(ctx, F1P_EINVAL, "E must be >= 0");
    if (!need_outputs_ok) return set_error(ctx, F1P_EINVAL, "steer, speed and best_idx outputs are required");
    if (cfg->n_stations < 2 || cfg->n_stations > 1024) return set_error(ctx, F1P_EINVAL, "n_stations must be in [2, 1024]");
    if (cfg->n_lookahead < 1 || cfg->n_lookahead > F1P_MAX_LOOKAHEADS || cfg->n_width < 1 || cfg->n_width > F1P_MAX_WIDTHS)
        return set_error(ctx, F1P_EINVAL, "n_lookahead / n_width out of range [1, 64]");
    if (cfg->n_shift < 0 || cfg->n_cull < 0) return set_error(ctx, F1P_EINVAL, "n_shift and n_cull must be >= 0");
    const int C = cfg->n_lookahead * cfg->n_width;
    if (cfg->cand_begin < 0 || cfg->cand_count < 0 || cfg->cand_begin + cfg->cand_count > C || (cfg->cand_count == 0 && cfg->cand_begin != 0))
        return set_error(ctx, F1P_EINVAL, "candidate shard [cand_begin, cand_begin+cand_count) outside [0, C)");
    if (cfg->generator != F1P_GEN_CLOTHOID && cfg->generator != F1P_GEN_CUBIC) return set_error(ctx, F1P_EINVAL, "unknown trajectory generator");
    if (ctx->n_wp < 2) return set_error(ctx, F1P_ESTATE, "waypoints not set: call f1p_set_waypoints first");
    if (device_goals && !ctx->has_psi) return set_error(ctx, F1P_ESTATE, "device goal sampling needs a heading column (col_psi >= 0)");
    return F1P_OK;
}

static int validate_kmpc(f1p_ctx* ctx, const f1p_kmpc_cfg* cfg, int E) {
    if (!cfg) return set_error(ctx, F1P_EINVAL, "cfg is NULL");
    if (E < 0) return set_error(ctx, F1P_EINVAL, "E must be >= 0");
    if (cfg->horizon < 1 || cfg->horizon > 4096) return set_error(ctx, F1P_EINVAL, "horizon must be in [1, 4096]");
    if (cfg->n_rollouts < 1) return set_error(ctx, F1P_EINVAL, "n_rollouts must be >= 1");
    if (!(cfg->dt > 0) || !(cfg->wheelbase > 0)) return set_error(ctx, F1P_EINVAL, "dt and wheelbase must be > 0");
    return F1P_OK;
}

}  // namespace f1p

using namespace f1p;

// ---------------------------------------------------------------------------------------------------
// RCCL, loaded with dlopen so libf1p.so itself has no link-time dependency on it.  Types and enum values come
// from <rccl/rccl.h> (ncclUint64, ncclInt32, ncclMin ...); only the entry points are resolved at run time.
// ---------------------------------------------------------------------------------------------------
static_assert(NCCL_UNIQUE_ID_BYTES == F1P_COMM_ID_BYTES, "f1p.h must carry RCCL's unique-id size");
typedef ncclResult_t (*pfn_ncclGetUniqueId)(ncclUniqueId*);
typedef ncclResult_t (*pfn_ncclCommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
typedef ncclResult_t (*pfn_ncclCommDestroy)(ncclComm_t);
typedef ncclResult_t (*pfn_ncclCommCount)(const ncclComm_t, int*);
typedef ncclResult_t (*pfn_ncclCommUserRank)(const ncclComm_t, int*);
typedef ncclResult_t (*pfn_ncclAllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
typedef ncclResult_t (*pfn_ncclAllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
typedef const char* (*pfn_ncclGetErrorString)(ncclResult_t);

static int rccl_open(f1p_ctx* ctx) {
    if (ctx->rccl_lib) return F1P_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        ctx->rccl_lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (ctx->rccl_lib) return F1P_OK;
    }
    return set_error(ctx, F1P_ECOMM, std::string("cannot load librccl.so: ") + dlerror());
}

template <typename F> static F rccl_sym(f1p_ctx* ctx, const char* name) { return (F)dlsym(ctx->rccl_lib, name); }

#pragma GCC visibility push(default)
extern "C" {

void f1p_lattice_cfg_default(f1p_lattice_cfg* cfg) {
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->n_stations = 100;                                   // lattice_planner.py:197
    cfg->n_lookahead = 4;                                    // :228
    cfg->n_width = 7;                                        // :229
    const double la[4] = {0.4, 0.6, 0.8, 1.0};
    for (int i = 0; i < 4; ++i) cfg->lookahead[i] = la[i];
    for (int i = 0; i < 7; ++i) cfg->width[i] = -1.0 + (2.0 / 6.0) * i;   // np.linspace(-1, 1, 7)
    cfg->width[6] = 1.0;
    cfg->n_shift = 1; cfg->n_cull = 1;
    cfg->check_collision = 1;
    cfg->w_length = 1.0;                                     // the only example cost that runs (:268-271)
    cfg->track_lookahead = 0.8;                              // :211
    cfg->wheelbase = 0.33;                                   // :55 (tracker default)
    cfg->max_reacquire = 20.0;                               // pure_pursuit.py:52
}

void f1p_kmpc_cfg_default(f1p_kmpc_cfg* cfg) {
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->horizon = 8; cfg->n_rollouts = 512;
    cfg->dt = 0.1; cfg->wheelbase = 0.33; cfg->max_steer = 0.4189; cfg->max_dsteer = 3.141592653589793;
    cfg->max_speed = 6.0; cfg->min_speed = 0.0; cfg->max_accel = 3.0;
    const double q[4] = {13.5, 13.5, 5.5, 13.0};
    for (int i = 0; i < 4; ++i) { cfg->q[i] = q[i]; cfg->qf[i] = q[i]; }
    cfg->r[0] = 0.01; cfg->r[1] = 100.0; cfg->rd[0] = 0.01; cfg->rd[1] = 100.0;
}

const char* f1p_version(void) { return F1P_VERSION_STRING; }

int f1p_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); return F1P_ENODEV; }
    return n;
}

int f1p_create(f1p_ctx** out, int device) {
    if (!out) return set_error(nullptr, F1P_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_error(nullptr, F1P_ENODEV, std::string("no HIP device visible (") + hipGetErrorString(e) + "): libf1p.so has no CPU fallback");
    if (device < 0 || device >= n) return set_error(nullptr, F1P_ENODEV, "device index out of range");
    f1p_ctx* ctx = new (std::nothrow) f1p_ctx();
    if (!ctx) return set_error(nullptr, F1P_ENOMEM, "out of host memory");
    ctx->device = device;
    int rc = F1P_OK;
    do {
        if ((e = hipSetDevice(device)) != hipSuccess) { rc = check_hip(nullptr, e, "hipSetDevice"); break; }
        if ((e = hipGetDeviceProperties(&ctx->prop, device)) != hipSuccess) { rc = check_hip(nullptr, e, "hipGetDeviceProperties"); break; }
        if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) { rc = check_hip(nullptr, e, "hipStreamCreate"); break; }
        if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) { rc = check_hip(nullptr, e, "hipEventCreate"); break; }
        if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) { rc = check_hip(nullptr, e, "hipEventCreate"); break; }
    } while (0);
    if (rc != F1P_OK) { f1p_destroy(ctx); return rc; }
    *out = ctx;
    return F1P_OK;
}

void f1p_destroy(f1p_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    f1p_comm_destroy(ctx);
    void* ptrs[] = {ctx->d_wx, ctx->d_wy, ctx->d_wv, ctx->d_wpsi, ctx->d_wkappa, ctx->d_wbox, ctx->d_bits, ctx->d_bits0, ctx->d_bits_clear, ctx->d_bb_scratch, ctx->d_arena, ctx->d_comm_key, ctx->d_comm_idx, ctx->d_kmpc_warm, ctx->d_kmpc_scratch, ctx->d_mix_scratch, ctx->d_split_scratch, ctx->d_rec_scratch, ctx->d_st_scratch, ctx->d_audit, ctx->d_audit_buf, ctx->d_cl_theta[0], ctx->d_cl_theta[1], ctx->d_step, ctx->d_comm_rec, ctx->d_order, ctx->d_kmpc_cfg};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
    if (ctx->h_step) (void)hipHostFree(ctx->h_step);
    if (ctx->h_kmpc_cfg) (void)hipHostFree(ctx->h_kmpc_cfg);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    for (auto& ev : ctx->ev_chunk) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : ctx->ev_prof) if (ev) (void)hipEventDestroy(ev);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (auto& st : ctx->pipe_stream) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (auto& ev : ctx->ev_pipe) if (ev) (void)hipEventDestroy(ev);
    if (ctx->rccl_lib) dlclose(ctx->rccl_lib);
    delete ctx;
}

const char* f1p_last_error(const f1p_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int f1p_device_info(const f1p_ctx* ctx, char* name, size_t name_len, int32_t* compute_units, char* arch, size_t arch_len) {
    if (!ctx) return F1P_EINVAL;
    if (name && name_len) { strncpy(name, ctx->prop.name, name_len - 1); name[name_len - 1] = 0; }
    if (compute_units) *compute_units = ctx->prop.multiProcessorCount;
    if (arch && arch_len) { strncpy(arch, ctx->prop.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    return F1P_OK;
}

#define F1P_ENTER(ctx)                                              \
    if (!(ctx)) return F1P_EINVAL;                                  \
    F1P_HIP((ctx), hipSetDevice((ctx)->device))

int f1p_dev_alloc(f1p_ctx* ctx, void** dptr, size_t bytes) {
    F1P_ENTER(ctx);
    if (!dptr) return set_error(ctx, F1P_EINVAL, "dptr is NULL");
    *dptr = nullptr;
    F1P_HIP(ctx, hipMalloc(dptr, bytes ? bytes : 1));
    return F1P_OK;
}
int f1p_dev_free(f1p_ctx* ctx, void* dptr) {
    F1P_ENTER(ctx);
    if (dptr) F1P_HIP(ctx, hipFree(dptr));
    return F1P_OK;
}
int f1p_host_alloc(f1p_ctx* ctx, void** hptr, size_t bytes) {
    F1P_ENTER(ctx);
    if (!hptr) return set_error(ctx, F1P_EINVAL, "hptr is NULL");
    *hptr = nullptr;
    F1P_HIP(ctx, hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, *hptr, 0) == hipSuccess && dev) ctx->host_blocks.push_back({(char*)*hptr, bytes ? bytes : 1, (char*)dev});
    else (void)hipGetLastError();                                // (not in the table: the per-call checks decide)
    return F1P_OK;
}
int f1p_host_free(f1p_ctx* ctx, void* hptr) {
    F1P_ENTER(ctx);
    if (hptr) {
        F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));             // a kernel may still be writing into it (zero-copy outputs).  FIRST: a failed sync returns with the block still tracked
        for (size_t i = 0; i < ctx->host_blocks.size(); ++i)
            if (ctx->host_blocks[i].base == (char*)hptr) { ctx->host_blocks.erase(ctx->host_blocks.begin() + (long)i); break; }
        F1P_HIP(ctx, hipHostFree(hptr));
    }
    return F1P_OK;
}
int f1p_h2d(f1p_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    F1P_ENTER(ctx);
    if (bytes && (!dst_dev || !src_host)) return set_error(ctx, F1P_EINVAL, "NULL pointer");
    if (bytes) F1P_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return F1P_OK;
}
int f1p_d2h(f1p_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    F1P_ENTER(ctx);
    if (bytes && (!dst_host || !src_dev)) return set_error(ctx, F1P_EINVAL, "NULL pointer");
    if (bytes) F1P_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return F1P_OK;
}
int f1p_memset(f1p_ctx* ctx, void* dst_dev, int value, size_t bytes) {
    F1P_ENTER(ctx);
    if (bytes) F1P_HIP(ctx, hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
    return F1P_OK;
}
int f1p_sync(f1p_ctx* ctx) {
    F1P_ENTER(ctx);
    F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F1P_OK;
}
int f1p_timer_begin(f1p_ctx* ctx) {
    F1P_ENTER(ctx);
    F1P_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return F1P_OK;
}
int f1p_timer_end(f1p_ctx* ctx, float* elapsed_ms) {
    F1P_ENTER(ctx);
    F1P_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    F1P_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    F1P_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    return F1P_OK;
}

// ---------------------------------------------------------------------------------------------------
int f1p_set_waypoints(f1p_ctx* ctx, const double* wp, int32_t n, int32_t ncols, int32_t col_x, int32_t col_y,
                      int32_t col_v, int32_t col_psi) {
    return f1p_set_waypoints_ex(ctx, wp, n, ncols, col_x, col_y, col_v, col_psi, -1);
}

int f1p_set_waypoints_ex(f1p_ctx* ctx, const double* wp, int32_t n, int32_t ncols, int32_t col_x, int32_t col_y,
                         int32_t col_v, int32_t col_psi, int32_t col_kappa) {
    F1P_ENTER(ctx);
    if (!wp) return set_error(ctx, F1P_EINVAL, "waypoints pointer is NULL");
    if (n < 2) return set_error(ctx, F1P_EINVAL, "at least 2 waypoints are required");
    if (ncols < 3) return set_error(ctx, F1P_EINVAL, "Waypoints needs to be a (Nxm), m >= 3, numpy array!");   // pure_pursuit.py:101-102
    auto bad = [&](int c) { return c < 0 || c >= ncols; };
    if (bad(col_x) || bad(col_y) || bad(col_v) || (col_psi >= 0 && bad(col_psi)) || (col_kappa >= 0 && bad(col_kappa)))
        return set_error(ctx, F1P_EINVAL, "column index out of range");
    std::vector<double> soa((size_t)5 * n, 0.0);
    if (col_psi >= 0)
        for (int i = 0; i < n; ++i) {
            const double psi = wp[(size_t)i * ncols + col_psi];
            if (!(psi >= -1.0e4 && psi <= 1.0e4)) return set_error(ctx, F1P_EINVAL, "waypoint heading must be finite and within +-1e4 rad");
        }
    for (int i = 0; i < n; ++i) {
        soa[i] = wp[(size_t)i * ncols + col_x];
        soa[(size_t)n + i] = wp[(size_t)i * ncols + col_y];
        soa[(size_t)2 * n + i] = wp[(size_t)i * ncols + col_v];
        soa[(size_t)3 * n + i] = col_psi >= 0 ? wp[(size_t)i * ncols + col_psi] : 0.0;
        soa[(size_t)4 * n + i] = col_kappa >= 0 ? wp[(size_t)i * ncols + col_kappa] : 0.0;
    }
    // bounding box of every 64-segment chunk (nearest_scan_boxed); infinite = "never skip this chunk"
    const int nchunk = (n - 1 + 63) / 64;
    std::vector<double> box((size_t)4 * nchunk);
    for (int c = 0; c < nchunk; ++c) {
        const int lo = 64 * c, hi = std::min(64 * c + 64, n - 1);
        double xmin = HUGE_VAL, xmax = -HUGE_VAL, ymin = HUGE_VAL, ymax = -HUGE_VAL;
        bool open_box = false;
        for (int i = lo; i <= hi; ++i) {
            const double x = soa[i], y = soa[(size_t)n + i];
            if (!(fabs(x) <= 1.0e6) || !(fabs(y) <= 1.0e6)) open_box = true;
            if (i < hi) {
                const double dx = soa[i + 1] - x, dy = soa[(size_t)n + i + 1] - y;
                if (!(dx * dx + dy * dy >= 1e-300)) open_box = true;   // zero-length (0/0 = NaN wins np.argmin) or NaN
            }
            xmin = std::min(xmin, x); xmax = std::max(xmax, x); ymin = std::min(ymin, y); ymax = std::max(ymax, y);
        }
        if (open_box) { xmin = ymin = -HUGE_VAL; xmax = ymax = HUGE_VAL; }
        box[4 * (size_t)c] = xmin; box[4 * (size_t)c + 1] = xmax; box[4 * (size_t)c + 2] = ymin; box[4 * (size_t)c + 3] = ymax;
    }
    F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n != ctx->n_wp) {
        double** ps[] = {&ctx->d_wx, &ctx->d_wy, &ctx->d_wv, &ctx->d_wpsi, &ctx->d_wkappa, &ctx->d_wbox};
        for (double** p : ps) { if (*p) (void)hipFree(*p); *p = nullptr; }
        ctx->n_wp = 0;
        for (double** p : ps)
            F1P_HIP(ctx, hipMalloc((void**)p, sizeof(double) * (p == &ctx->d_wbox ? (size_t)4 * nchunk : (size_t)n)));
    }
    F1P_HIP(ctx, hipMemcpy(ctx->d_wbox, box.data(), sizeof(double) * box.size(), hipMemcpyHostToDevice));
    const size_t b = sizeof(double) * (size_t)n;
    F1P_HIP(ctx, hipMemcpy(ctx->d_wx, soa.data(), b, hipMemcpyHostToDevice));
    F1P_HIP(ctx, hipMemcpy(ctx->d_wy, soa.data() + n, b, hipMemcpyHostToDevice));
    F1P_HIP(ctx, hipMemcpy(ctx->d_wv, soa.data() + 2 * (size_t)n, b, hipMemcpyHostToDevice));
    F1P_HIP(ctx, hipMemcpy(ctx->d_wpsi, soa.data() + 3 * (size_t)n, b, hipMemcpyHostToDevice));
    F1P_HIP(ctx, hipMemcpy(ctx->d_wkappa, soa.data() + 4 * (size_t)n, b, hipMemcpyHostToDevice));
    ctx->n_wp = n;
    ctx->has_psi = col_psi >= 0;
    ctx->has_kappa = col_kappa >= 0;
    return F1P_OK;
}

static void drop_grid(f1p_ctx* ctx) {
    if (ctx->d_bits) (void)hipFree(ctx->d_bits);
    if (ctx->d_bits0) (void)hipFree(ctx->d_bits0);
    if (ctx->d_bits_clear) (void)hipFree(ctx->d_bits_clear);
    ctx->d_bits_clear = nullptr; ctx->clear_dist = 0.0;
    ctx->d_bits = nullptr; ctx->d_bits0 = nullptr; ctx->has_grid = false; ctx->inflate_radius = 0.0; ctx->user_inflate = 0.0; ctx->disc_radius = 0.0; ctx->n_disc = 0;
}

int f1p_set_grid(f1p_ctx* ctx, const uint8_t* img, int32_t w, int32_t h, double res, double ox, double oy,
                 int32_t occupied_below) {
    F1P_ENTER(ctx);
    if (!img) {   // clear
        F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
        drop_grid(ctx);
        return F1P_OK;
    }
    if (w < 1 || h < 1 || w > 65535 || h > 65535) return set_error(ctx, F1P_EINVAL, "grid size out of range");
    if (!(res > 0.0)) return set_error(ctx, F1P_EINVAL, "resolution must be > 0");
    F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drop_grid(ctx);
    ctx->gw = w; ctx->gh = h; ctx->gwwords = (w + 31) / 32;
    ctx->res = res; ctx->inv_res = 1.0 / res; ctx->ox = ox; ctx->oy = oy;
    const size_t bit_bytes = sizeof(uint32_t) * (size_t)ctx->gwwords * h;
    uint8_t* d_img = nullptr;
    F1P_HIP(ctx, hipMalloc((void**)&d_img, (size_t)w * h));
    int rc = check_hip(ctx, hipMalloc((void**)&ctx->d_bits0, bit_bytes), "hipMalloc(bits)");
    if (rc == F1P_OK) rc = check_hip(ctx, hipMalloc((void**)&ctx->d_bits, bit_bytes), "hipMalloc(bits)");
    if (rc == F1P_OK) rc = check_hip(ctx, hipMemcpyAsync(d_img, img, (size_t)w * h, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(img)");
    if (rc == F1P_OK) rc = launch_pack_grid(ctx, d_img, w, h, occupied_below);
    if (rc == F1P_OK) rc = check_hip(ctx, hipMemcpyAsync(ctx->d_bits, ctx->d_bits0, bit_bytes, hipMemcpyDeviceToDevice, ctx->stream), "hipMemcpyAsync(bits)");
    if (rc == F1P_OK) rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    (void)hipFree(d_img);
    if (rc != F1P_OK) { drop_grid(ctx); return rc; }
    ctx->has_grid = true;
    return F1P_OK;
}

static int edt_cap_check(f1p_ctx* ctx, int64_t cap) {
    if (!ctx->has_grid) return set_error(ctx, F1P_ESTATE, "occupancy grid not set: call f1p_set_grid first");
    if (cap < 1 || cap > 8192) return set_error(ctx, F1P_EINVAL, "distance cap must be within 1..8192 cells");
    return F1P_OK;
}

int f1p_grid_distance_batch(f1p_ctx* ctx, float* dist, int32_t cap_cells) {
    F1P_ENTER(ctx);
    if (!dist) return set_error(ctx, F1P_EINVAL, "dist is NULL");
    int rc = edt_cap_check(ctx, cap_cells);
    if (rc) return rc;
    const size_t n = (size_t)ctx->gw * ctx->gh;
    float* d_dist = nullptr;
    F1P_HIP(ctx, hipMalloc((void**)&d_dist, sizeof(float) * n));
    rc = launch_grid_edt(ctx, cap_cells, 0u, d_dist, nullptr, nullptr);
    if (rc == F1P_OK) rc = check_hip(ctx, hipMemcpy(dist, d_dist, sizeof(float) * n, hipMemcpyDeviceToHost), "hipMemcpy(dist)");
    (void)hipFree(d_dist);
    return rc;
}

// the active bitmap = the uploaded grid dilated by ONE disc of radius user_inflate + disc_radius (exact Euclidean distance
// transform, k_grid.hip): the user's inflation and the footprint's disc radius add, neither replaces the other
static int apply_dilation(f1p_ctx* ctx, double radius) {
    F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->clear_dist = 0.0;                                  // the active bitmap changes: its clearance map is stale
    if (radius == 0.0) {
        F1P_HIP(ctx, hipMemcpy(ctx->d_bits, ctx->d_bits0, sizeof(uint32_t) * (size_t)ctx->gwwords * ctx->gh, hipMemcpyDeviceToDevice));
        ctx->inflate_radius = 0.0;
        return F1P_OK;
    }
    const double q = radius * ctx->inv_res;
    const double thr = ceil(q * q);                       // integer d2 < q^2  <=>  d2 < ceil(q^2)
    const int64_t cap = (int64_t)ceil(q) + 1;
    int rc = edt_cap_check(ctx, cap);
    if (rc) return rc;
    rc = launch_grid_edt(ctx, (int)cap, (uint32_t)thr, nullptr, nullptr, ctx->d_bits);
    if (rc == F1P_OK) ctx->inflate_radius = radius;
    return rc;
}

int f1p_inflate_grid(f1p_ctx* ctx, double radius) {
    F1P_ENTER(ctx);
    if (!ctx->has_grid) return set_error(ctx, F1P_ESTATE, "occupancy grid not set: call f1p_set_grid first");
    if (!(radius >= 0.0) || !isfinite(radius)) return set_error(ctx, F1P_EINVAL, "inflation radius must be finite and >= 0");
    const int rc = apply_dilation(ctx, radius + ctx->disc_radius);
    if (rc == F1P_OK) ctx->user_inflate = radius;
    return rc;
}

int f1p_set_footprint(f1p_ctx* ctx, int32_t n_discs, const double* offsets, double radius) {
    F1P_ENTER(ctx);
    if (n_discs < 0 || n_discs > 4 || (n_discs > 0 && !offsets)) return set_error(ctx, F1P_EINVAL, "between 0 and 4 footprint discs are supported");
    if (!ctx->has_grid) return set_error(ctx, F1P_ESTATE, "occupancy grid not set: call f1p_set_grid first");
    if (n_discs > 0 && (!(radius >= 0.0) || !isfinite(radius))) return set_error(ctx, F1P_EINVAL, "footprint disc radius must be finite and >= 0");
    for (int d = 0; d < n_discs; ++d)
        if (!isfinite(offsets[d]) || fabs(offsets[d]) > 100.0) return set_error(ctx, F1P_EINVAL, "footprint offsets must be finite and within +-100 m");
    const double disc = n_discs > 0 ? radius : 0.0;
    int rc = apply_dilation(ctx, ctx->user_inflate + disc);           // the disc radius adds to the user's inflation; n_discs = 0 restores it
    if (rc) return rc;
    ctx->disc_radius = disc;
    ctx->n_disc = n_discs;
    for (int d = 0; d < 4; ++d) ctx->disc_off[d] = d < n_discs ? offsets[d] : 0.0;
    return F1P_OK;
}

// ---------------------------------------------------------------------------------------------------
int f1p_nearest_point_batch(f1p_ctx* ctx, const double* pts, int32_t E, double* proj, double* dist, double* t, int32_t* idx) {
    F1P_ENTER(ctx);
    if (E < 0 || (E > 0 && !pts)) return set_error(ctx, F1P_EINVAL, "bad pts / E");
    if (ctx->n_wp < 2) return set_error(ctx, F1P_ESTATE, "waypoints not set");
    Stage s(ctx);
    s.need(sizeof(double) * 2 * E); s.need(sizeof(double) * 2 * E, proj); s.need(sizeof(double) * E, dist);
    s.need(sizeof(double) * E, t); s.need(sizeof(int32_t) * E, idx);
    int rc = s.begin(); if (rc) return rc;
    const double* d_pts;
    if ((rc = s.in(pts, (size_t)2 * E, &d_pts))) return rc;
    double* d_proj = s.out(proj, (size_t)2 * E); double* d_dist = s.out(dist, E); double* d_t = s.out(t, E);
    int32_t* d_idx = s.out(idx, E);
    if ((rc = launch_nearest(ctx, d_pts, E, d_proj, d_dist, d_t, d_idx))) return rc;
    return s.finish();
}

int f1p_intersect_point_batch(f1p_ctx* ctx, const double* pts, const double* start_t, int32_t E, double radius,
                              int32_t wrap, double* first_p, int32_t* first_i, double* first_t, int32_t* found) {
    F1P_ENTER(ctx);
    if (E < 0 || (E > 0 && (!pts || !start_t))) return set_error(ctx, F1P_EINVAL, "bad pts / start_t / E");
    if (ctx->n_wp < 2) return set_error(ctx, F1P_ESTATE, "waypoints not set");
    for (int i = 0; i < E; ++i)
        if (!(start_t[i] >= 0.0) || !(start_t[i] <= (double)ctx->n_wp)) return set_error(ctx, F1P_EINVAL, "start_t must be in [0, n]");
    Stage s(ctx);
    s.need(sizeof(double) * 2 * E); s.need(sizeof(double) * E); s.need(sizeof(double) * 2 * E, first_p);
    s.need(sizeof(int32_t) * E, first_i); s.need(sizeof(double) * E, first_t); s.need(sizeof(int32_t) * E, found);
    int rc = s.begin(); if (rc) return rc;
    const double *d_pts, *d_st;
    if ((rc = s.in(pts, (size_t)2 * E, &d_pts))) return rc;
    if ((rc = s.in(start_t, (size_t)E, &d_st))) return rc;
    double* d_p = s.out(first_p, (size_t)2 * E); int32_t* d_i = s.out(first_i, E); double* d_t = s.out(first_t, E);
    int32_t* d_f = s.out(found, E);
    if ((rc = launch_intersect(ctx, d_pts, d_st, E, radius, wrap, d_p, d_i, d_t, d_f))) return rc;
    return s.finish();
}

int f1p_pure_pursuit_dev(f1p_ctx* ctx, const double* d_poses, int32_t E, double lookahead, double wheelbase,
                         double max_reacquire, double* d_steer, double* d_speed, int32_t* d_near_idx,
                         int32_t* d_la_idx, int32_t* d_status) {
    F1P_ENTER(ctx);
    if (E < 0 || (E > 0 && (!d_poses || !d_steer || !d_speed))) return set_error(ctx, F1P_EINVAL, "poses, steer and speed are required");
    if (ctx->n_wp < 2) return set_error(ctx, F1P_ESTATE, "Please set waypoints to track during planner instantiation or when calling plan()");
    return launch_pure_pursuit(ctx, d_poses, E, lookahead, wheelbase, max_reacquire, d_steer, d_speed, d_near_idx, d_la_idx, d_status);
}

int f1p_pure_pursuit_batch(f1p_ctx* ctx, const double* poses, int32_t E, double lookahead, double wheelbase,
                           double max_reacquire, double* steer, double* speed, int32_t* near_idx, int32_t* la_idx,
                           int32_t* status) {
    F1P_ENTER(ctx);
    if (E < 0 || (E > 0 && (!poses || !steer || !speed))) return set_error(ctx, F1P_EINVAL, "poses, steer and speed are required");
    Stage s(ctx);
    s.need(sizeof(double) * 3 * E); s.need(sizeof(double) * E); s.need(sizeof(double) * E);
    s.need(sizeof(int32_t) * E, near_idx); s.need(sizeof(int32_t) * E, la_idx); s.need(sizeof(int32_t) * E, status);
    int rc = s.begin(); if (rc) return rc;
    const double* d_poses;
    if ((rc = s.in(poses, (size_t)3 * E, &d_poses))) return rc;
    double* d_steer = s.out(steer, E); double* d_speed = s.out(speed, E);
    int32_t* d_n = s.out(near_idx, E); int32_t* d_l = s.out(la_idx, E); int32_t* d_s = s.out(status, E);
    if ((rc = f1p_pure_pursuit_dev(ctx, d_poses, E, lookahead, wheelbase, max_reacquire, d_steer, d_speed, d_n, d_l, d_s))) return rc;
    return s.finish();
}

// ---------------------------------------------------------------------------------------------------
int f1p_stanley_batch(f1p_ctx* ctx, const double* states, int32_t E, double wheelbase, double k_path, double* steer,
                      double* speed, int32_t* near_idx) {
    F1P_ENTER(ctx);
    if (E < 0 || (E > 0 && (!states || !steer || !speed))) return set_error(ctx, F1P_EINVAL, "states, steer and speed are required");
    if (ctx->n_wp < 2) return set_error(ctx, F1P_ESTATE, "Please set waypoints to track during planner instantiation or when calling plan()");
    if (!ctx->has_psi) return set_error(ctx, F1P_EINVAL, "Waypoints needs to be a (Nxm), m >= 4, numpy array!");   // stanley.py:131-132
    Stage s(ctx);
    s.need(8 * 4 * (size_t)E); s.need(8 * (size_t)E); s.need(8 * (size_t)E); s.need(4 * (size_t)E, near_idx);
    int rc = s.begin(); if (rc) return rc;
    const double* d_st;
    if ((rc = s.in(states, (size_t)4 * E, &d_st))) return rc;
    double* d_steer = s.out(steer, E); double* d_speed = s.out(speed, E); int32_t* d_n = s.out(near_idx, E);
    if ((rc = launch_stanley(ctx, d_st, E, wheelbase, k_path, d_steer, d_speed, d_n))) return rc;
    return s.finish();
}

int f1p_lqr_batch(f1p_ctx* ctx, const double* states, double* err, int32_t E, double wheelbase, double timestep,
                  const double q[4], double r, int32_t max_iter, double eps, double* steer, double* speed,
                  int32_t* near_idx) {
    F1P_ENTER(ctx);
    if (E < 0 || (E > 0 && (!states || !err || !steer || !speed)) || !q) return set_error(ctx, F1P_EINVAL, "states, err, q, steer and speed are required");
    if (!(timestep > 0.0) || !(wheelbase > 0.0) || max_iter < 0) return set_error(ctx, F1P_EINVAL, "timestep and wheelbase must be > 0, max_iter >= 0");
    if (ctx->n_wp < 2) return set_error(ctx, F1P_ESTATE, "Please set waypoints to track during planner instantiation or when calling plan()");
    if (!ctx->has_psi || !ctx->has_kappa) return set_error(ctx, F1P_EINVAL, "Waypoints needs to be a (Nxm), m >= 5, numpy array!");   // lqr.py:195-196
    Stage s(ctx);
    s.need(8 * 4 * (size_t)E); s.need(8 * 2 * (size_t)E); s.need(8 * (size_t)E); s.need(8 * (size_t)E); s.need(4 * (size_t)E, near_idx);
    int rc = s.begin(); if (rc) return rc;
    const double* d_st;
    if ((rc = s.in(states, (size_t)4 * E, &d_st))) return rc;
    const double* d_err_in;
    if ((rc = s.in((const double*)err, (size_t)2 * E, &d_err_in))) return rc;
    double* d_err = const_cast<double*>(d_err_in);
    if (E > 0) s.outs.push_back({(void*)err, (void*)d_err, sizeof(double) * 2 * (size_t)E});   // in/out
    double* d_steer = s.out(steer, E); double* d_speed = s.out(speed, E); int32_t* d_n = s.out(near_idx, E);
    if ((rc = launch_lqr(ctx, d_st, d_err, E, wheelbase, timestep, q, r, max_iter, eps, d_steer, d_speed, d_n))) return rc;
    return s.finish();
}

// ---------------------------------------------------------------------------------------------------
// Closed-loop mode (f1p_lattice_set_closed_loop): every plan leaves its winners' heading column on the device, [E][S] fp64 in one of two
// ctx-owned buffers used alternately, and the next plan of the same shape that passes prev_theta == NULL takes it as its previous path
// (get_similarity_cost, lattice_planner.py:287-296: the reference compares with the previous plan's best trajectory).  Nothing crosses PCIe.
struct ClosedLoop {
    const double* prev = nullptr;      // the plan's prev_theta (the caller's, or the kept headings, or null = first plan)
    double* out = nullptr;             // where its winners' headings go (null: closed loop off)
    bool from_ctx = false;
};

static int cl_begin(f1p_ctx* ctx, const double* d_prev_caller, int E, int S, bool writes, ClosedLoop* cl) {
    cl->prev = d_prev_caller; cl->out = nullptr; cl->from_ctx = false;
    if (!ctx->lattice_closed_loop || E <= 0) return F1P_OK;
    const size_t need = sizeof(double) * (size_t)E * S;
    if (need > ctx->cl_bytes) {
        F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (auto& b : ctx->d_cl_theta) { if (b) (void)hipFree(b); b = nullptr; }
        ctx->cl_bytes = 0; ctx->cl_valid = false;
        for (auto& b : ctx->d_cl_theta) {
            if (hipMalloc((void**)&b, need) != hipSuccess) {         // both buffers or none (a failed second allocation left the first behind with cl_bytes = 0)
                (void)hipGetLastError();
                for (auto& q : ctx->d_cl_theta) { if (q) (void)hipFree(q); q = nullptr; }
                return set_error(ctx, F1P_ENOMEM, "closed-loop heading buffers: out of device memory");
            }
        }
        ctx->cl_bytes = need;
    }
    if (ctx->cl_valid && (ctx->cl_E != E || ctx->cl_S != S)) ctx->cl_valid = false;   // another batch shape: a first plan again
    if (!d_prev_caller && ctx->cl_valid) { cl->prev = ctx->d_cl_theta[ctx->cl_cur]; cl->from_ctx = true; }
    if (writes) cl->out = ctx->d_cl_theta[ctx->cl_valid ? ctx->cl_cur ^ 1 : 0];
    return F1P_OK;
}

static void cl_commit(f1p_ctx* ctx, const ClosedLoop& cl, int E, int S) {
    if (!cl.out) return;
    ctx->cl_cur = cl.out == ctx->d_cl_theta[0] ? 0 : 1;
    ctx->cl_valid = true; ctx->cl_E = E; ctx->cl_S = S;
}

static int lattice_plan_dev_impl(f1p_ctx* ctx, const double* d_poses, const double* d_goals, const double* d_prev_theta,
                                 int32_t E, const f1p_lattice_cfg* cfg, double* d_steer, double* d_speed,
                                 int32_t* d_best_idx, double* d_best_cost, int32_t* d_status, int32_t* d_near_idx,
                                 double* d_best_traj, double* d_all_cost, double* d_all_traj, float* d_best_traj32, double* d_theta_out = nullptr,
                                 double* d_pose_copy = nullptr) {
    int rc = validate_lattice(ctx, cfg, E, d_goals == nullptr, E == 0 || (d_poses && d_best_idx && (cfg && cfg->cand_count > 0 ? true : (d_steer && d_speed))));
    if (rc) return rc;
    // a candidate shard evaluates only (cost + index); the emit half runs after the cross-rank argmin
    const int mode = cfg->cand_count > 0 ? LATTICE_EVAL : LATTICE_FULL;
    return launch_lattice(ctx, mode, d_poses, d_goals, d_prev_theta, E, cfg, nullptr, nullptr, d_steer, d_speed, d_best_idx,
                          d_best_cost, d_status, d_near_idx, d_best_traj, d_all_cost, d_all_traj, d_best_traj32, d_theta_out, d_pose_copy);
}

// the *_dev entry points in closed-loop mode: prev_theta == NULL means "the headings the previous plan left on the device"
static int lattice_plan_dev_cl(f1p_ctx* ctx, const double* d_poses, const double* d_goals, const double* d_prev_theta,
                               int32_t E, const f1p_lattice_cfg* cfg, double* d_steer, double* d_speed,
                               int32_t* d_best_idx, double* d_best_cost, int32_t* d_status, int32_t* d_near_idx,
                               double* d_best_traj, double* d_all_cost, double* d_all_traj, float* d_best_traj32) {
    ClosedLoop cl;
    // the configuration is validated BEFORE cl_begin touches the kept headings (an invalid cfg -- a huge n_stations -- used to free
    // them and fail in hipMalloc instead of returning F1P_EINVAL); lattice_plan_dev_impl validates again, with the output pointers
    if (const int rcv = validate_lattice(ctx, cfg, E, d_goals == nullptr, true)) return rcv;
    if (cfg && E > 0 && cfg->n_stations >= 2) {
        const int rc0 = cl_begin(ctx, d_prev_theta, E, cfg->n_stations, cfg->cand_count == 0, &cl);
        if (rc0) return rc0;
    } else cl.prev = d_prev_theta;
    const int rc = lattice_plan_dev_impl(ctx, d_poses, d_goals, cl.prev, E, cfg, d_steer, d_speed, d_best_idx, d_best_cost, d_status, d_near_idx,
                                         d_best_traj, d_all_cost, d_all_traj, d_best_traj32, cl.out);
    if (rc == F1P_OK) cl_commit(ctx, cl, E, cfg->n_stations);
    return rc;
}

int f1p_lattice_plan_dev(f1p_ctx* ctx, const double* d_poses, const double* d_goals, const double* d_prev_theta,
                         int32_t E, const f1p_lattice_cfg* cfg, double* d_steer, double* d_speed,
                         int32_t* d_best_idx, double* d_best_cost, int32_t* d_status, int32_t* d_near_idx,
                         double* d_best_traj, double* d_all_cost, double* d_all_traj) {
    F1P_ENTER(ctx);
    return lattice_plan_dev_cl(ctx, d_poses, d_goals, d_prev_theta, E, cfg, d_steer, d_speed, d_best_idx, d_best_cost, d_status, d_near_idx,
                               d_best_traj, d_all_cost, d_all_traj, nullptr);
}

int f1p_lattice_plan_dev_f32(f1p_ctx* ctx, const double* d_poses, const double* d_goals, const double* d_prev_theta,
                             int32_t E, const f1p_lattice_cfg* cfg, double* d_steer, double* d_speed,
                             int32_t* d_best_idx, double* d_best_cost, int32_t* d_status, int32_t* d_near_idx, float* d_best_traj32) {
    F1P_ENTER(ctx);
    return lattice_plan_dev_cl(ctx, d_poses, d_goals, d_prev_theta, E, cfg, d_steer, d_speed, d_best_idx, d_best_cost, d_status, d_near_idx,
                               nullptr, nullptr, nullptr, d_best_traj32);
}

int f1p_lattice_emit_dev(f1p_ctx* ctx, const double* d_poses, const double* d_goals, int32_t E,
                         const f1p_lattice_cfg* cfg, const int32_t* d_cand_idx, const double* d_cand_cost,
                         double* d_steer, double* d_speed, int32_t* d_status, int32_t* d_near_idx, double* d_best_traj) {
    F1P_ENTER(ctx);
    int rc = validate_lattice(ctx, cfg, E, d_goals == nullptr, E == 0 || (d_poses && d_cand_idx && d_steer && d_speed));
    if (rc) return rc;
    ClosedLoop cl;                                                 // the emitted winners are this plan's previous path for the next one
    if (E > 0 && (rc = cl_begin(ctx, nullptr, E, cfg->n_stations, true, &cl))) return rc;
    rc = launch_lattice(ctx, LATTICE_EMIT, d_poses, d_goals, nullptr, E, cfg, d_cand_idx, d_cand_cost, d_steer, d_speed,
                        nullptr, nullptr, d_status, d_near_idx, d_best_traj, nullptr, nullptr, nullptr, cl.out);
    if (rc == F1P_OK) cl_commit(ctx, cl, E, cfg->n_stations);
    return rc;
}

int f1p_lattice_set_closed_loop(f1p_ctx* ctx, int32_t on) {
    if (!ctx) return F1P_EINVAL;
    ctx->lattice_closed_loop = on != 0;
    ctx->cl_valid = false;                                         // (re)armed: the next plan is a first plan
    ctx->step_chain = false;
    return F1P_OK;
}

int f1p_lattice_closed_loop_state(f1p_ctx* ctx, const double** d_prev_theta, int32_t* E, int32_t* S) {
    if (!ctx) return F1P_EINVAL;
    const bool v = (ctx->lattice_closed_loop || ctx->step_chain) && ctx->cl_valid;   // (the headings the next plan -- or, for a step chain, the next STEP -- would use)
    if (d_prev_theta) *d_prev_theta = v ? ctx->d_cl_theta[ctx->cl_cur] : nullptr;
    if (E) *E = v ? ctx->cl_E : 0;
    if (S) *S = v ? ctx->cl_S : 0;
    return F1P_OK;
}

// TRAJ = double (the reference's fp64 rows) or float (f1p_lattice_plan_batch_f32: the same rows rounded once on the device)
extern "C++" {
template <typename TRAJ>
static int lattice_plan_batch_impl(f1p_ctx* ctx, const double* poses, const double* goals, const double* prev_theta,
                                   int32_t E, const f1p_lattice_cfg* cfg, double* steer, double* speed, int32_t* best_idx,
                                   double* best_cost, int32_t* status, int32_t* near_idx, TRAJ* best_traj,
                                   double* all_cost, double* all_traj) {
    constexpr bool F32 = sizeof(TRAJ) == 4;
    int rc = validate_lattice(ctx, cfg, E, goals == nullptr, E == 0 || (poses && best_idx && ((cfg && cfg->cand_count > 0) || (steer && speed))));
    if (rc) return rc;
    if (cfg->cand_count > 0 && (steer || speed || status || best_traj))
        return set_error(ctx, F1P_EINVAL, "a candidate shard (cfg.cand_count > 0) only evaluates: it produces best_idx, best_cost and near_idx; "
                                          "pass NULL for steer / speed / status / best_traj and emit the global winner with f1p_lattice_emit_dev");
    const size_t C = (size_t)cfg->n_lookahead * cfg->n_width, S = cfg->n_stations, e = E;
    // (page-locked best_traj: see below -- decided here so that the arena does not reserve bytes nobody uses)
    TRAJ* bt_dev = (best_traj && !all_cost && !all_traj && E >= F1P_ZEROCOPY_MIN_EGOS)
                       ? (TRAJ*)pinned_device_ptr(ctx, best_traj, sizeof(TRAJ) * e * S * 4) : nullptr;
    const bool pinned = bt_dev != nullptr;
    const bool zero_copy_traj = F1P_TRAJ_ZEROCOPY && pinned && E < 8192;
    // Round 6: with the trajectories going straight into the caller's page-locked array, the poses and the per-ego result columns do too when
    // THEY are page-locked -- k_lattice_prologue reads the poses out of host memory and leaves a device copy (f1p_lattice_step_batch's scheme), the selection
    // kernel stores steer / speed / index / cost / status / nearest segment where the caller reads them: no hipMemcpy in either direction, each of which is
    // ~10 us of submission and DMA start-up in front of / behind a 65 us plan.  (All-fp64 mode keeps the copies: k_lattice re-reads the pose per thread.)
    const double* zp = nullptr; double *zs = nullptr, *zv = nullptr, *zc_ = nullptr; int32_t *zi = nullptr, *zt = nullptr, *zn = nullptr;
    bool zc_io = F1P_IO_ZEROCOPY && (best_traj ? zero_copy_traj : (E >= F1P_ZEROCOPY_MIN_EGOS && E < 8192 && !all_cost && !all_traj)) && !goals && !prev_theta &&
                 ctx->lattice_mixed != 0 && cfg->cand_count == 0;
    if (zc_io) {
        zp = (const double*)pinned_device_ptr(ctx, poses, 32 * e);
        zs = (double*)pinned_device_ptr(ctx, steer, 8 * e); zv = (double*)pinned_device_ptr(ctx, speed, 8 * e);
        zi = (int32_t*)pinned_device_ptr(ctx, best_idx, 4 * e);
        zc_ = best_cost ? (double*)pinned_device_ptr(ctx, best_cost, 8 * e) : nullptr;
        zt = status ? (int32_t*)pinned_device_ptr(ctx, status, 4 * e) : nullptr;
        zn = near_idx ? (int32_t*)pinned_device_ptr(ctx, near_idx, 4 * e) : nullptr;
        zc_io = zp && zs && zv && zi && (!best_cost || zc_) && (!status || zt) && (!near_idx || zn);
    }
    Stage s(ctx);
    if (zc_io) {
        s.need(8 * 4 * e);                                      // the device copy of the poses
        if ((rc = s.begin())) return rc;
        double* d_pose_copy = (double*)arena_take(ctx, 8 * 4 * e);
        ClosedLoop cl;
        if ((rc = cl_begin(ctx, nullptr, E, (int)S, true, &cl))) return rc;
        struct HostDst { f1p_ctx* c; ~HostDst() { c->traj_dst_host = false; } } host_dst{ctx};
        ctx->traj_dst_host = best_traj != nullptr;
        double* bt64 = nullptr; float* bt32 = nullptr;
        if (best_traj) { if (F32) bt32 = reinterpret_cast<float*>(bt_dev); else bt64 = reinterpret_cast<double*>(bt_dev); }
        if ((rc = lattice_plan_dev_impl(ctx, zp, nullptr, cl.prev, E, cfg, zs, zv, zi, zc_, zt, zn, bt64, nullptr, nullptr, bt32, cl.out, d_pose_copy))) return rc;
        cl_commit(ctx, cl, E, (int)S);
        F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return F1P_OK;
    }
    s.need(8 * 4 * e); s.need(8 * e * C * 3, goals); s.need(8 * e * S, prev_theta);
    s.need(8 * e, steer); s.need(8 * e, speed); s.need(4 * e); s.need(8 * e, best_cost); s.need(4 * e, status); s.need(4 * e, near_idx);
    s.need(sizeof(TRAJ) * e * S * 4, best_traj && !zero_copy_traj); s.need(8 * e * C, all_cost); s.need(8 * e * C * S * 4, all_traj);
    if ((rc = s.begin())) return rc;
    const double *d_poses, *d_goals, *d_prev;
    if ((rc = s.in(poses, 4 * e, &d_poses))) return rc;
    if ((rc = s.in(goals, e * C * 3, &d_goals))) return rc;
    if ((rc = s.in(prev_theta, e * S, &d_prev))) return rc;
    ClosedLoop cl;                                             // closed-loop mode: prev_theta == NULL = the headings the previous plan left on the device
    if (E > 0 && (rc = cl_begin(ctx, d_prev, E, (int)S, cfg->cand_count == 0, &cl))) return rc;
    d_prev = cl.prev;
    double* d_steer = s.out(steer, e); double* d_speed = s.out(speed, e); int32_t* d_bi = s.out(best_idx, e);
    double* d_bc = s.out(best_cost, e); int32_t* d_st = s.out(status, e); int32_t* d_ni = s.out(near_idx, e);
    // Trajectories into PAGE-LOCKED host memory (f1p_host_alloc / hipHostRegister), 2048 <= E < 8192: the selection kernel writes
    // them straight into the caller's array (it is device-visible), so the PCIe writes stream while the kernel still runs and no
    // copy is submitted at all -- measured at 4096 egos, p50 host to host: fp64 rows 0.277 ms (0.287 with one hipMemcpyAsync behind
    // the plan, 0.294 with the plan in two slices and the copies on a second stream, the round-2 scheme), f32 rows 0.214 (0.230 /
    // 0.233); slicing only the selection kernel was worse still (0.30 / 0.24: every cross-stream edge costs ~10 us).
    // From 8192 egos the batch is planned in slices of >= 4096 egos whose results travel on a second stream while the next slice
    // is planned (only into page-locked memory: copies into pageable memory block the calling thread and would serialise the slices).
    TRAJ* d_bt = zero_copy_traj ? bt_dev : s.out(best_traj, e * S * 4);
    struct HostDst { f1p_ctx* c; ~HostDst() { c->traj_dst_host = false; } } host_dst{ctx};   // (reset on every way out)
    ctx->traj_dst_host = zero_copy_traj;
    double* d_ac = s.out(all_cost, e * C); double* d_at = s.out(all_traj, e * C * S * 4);
    auto plan = [&](size_t e0, size_t n) {
        double* bt64 = nullptr; float* bt32 = nullptr;
        if (d_bt) { if (F32) bt32 = reinterpret_cast<float*>(d_bt) + e0 * S * 4; else bt64 = reinterpret_cast<double*>(d_bt) + e0 * S * 4; }
        return lattice_plan_dev_impl(ctx, d_poses + 4 * e0, d_goals ? d_goals + e0 * C * 3 : nullptr, d_prev ? d_prev + e0 * S : nullptr,
                                     (int32_t)n, cfg, d_steer ? d_steer + e0 : nullptr, d_speed ? d_speed + e0 : nullptr, d_bi + e0, d_bc ? d_bc + e0 : nullptr,
                                     d_st ? d_st + e0 : nullptr, d_ni ? d_ni + e0 : nullptr, bt64, e0 == 0 && n == e ? d_ac : nullptr,
                                     e0 == 0 && n == e ? d_at : nullptr, bt32, cl.out ? cl.out + e0 * S : nullptr);
    };
    const int K = (pinned && E >= 8192) ? (E / 4096 < F1P_PLAN_CHUNKS_MAX ? E / 4096 : F1P_PLAN_CHUNKS_MAX) : 1;
    if (K > 1 && (rc = ensure_copy_stream(ctx))) return rc;
    if (K == 1) {
        if ((rc = plan(0, e))) return rc;
        cl_commit(ctx, cl, E, (int)S);
        return s.finish();
    }
    for (int k = 0; k < K; ++k) {
        const size_t e0 = e * k / K, e1 = e * (k + 1) / K, n = e1 - e0;
        if ((rc = plan(e0, n))) return rc;
        F1P_HIP(ctx, hipEventRecord(ctx->ev_chunk[k], ctx->stream));
        F1P_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_chunk[k], 0));
        F1P_HIP(ctx, hipMemcpyAsync(best_traj + e0 * S * 4, d_bt + e0 * S * 4, n * S * 4 * sizeof(TRAJ), hipMemcpyDeviceToHost, ctx->copy_stream));
    }
    cl_commit(ctx, cl, E, (int)S);
    rc = s.gather(best_traj);                                  // the per-ego scalars of every slice: one copy + scatter, on the planning stream
    F1P_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    return rc;
}
}  // extern "C++"

int f1p_lattice_plan_batch(f1p_ctx* ctx, const double* poses, const double* goals, const double* prev_theta,
                           int32_t E, const f1p_lattice_cfg* cfg, double* steer, double* speed, int32_t* best_idx,
                           double* best_cost, int32_t* status, int32_t* near_idx, double* best_traj,
                           double* all_cost, double* all_traj) {
    F1P_ENTER(ctx);
    return lattice_plan_batch_impl<double>(ctx, poses, goals, prev_theta, E, cfg, steer, speed, best_idx, best_cost, status, near_idx, best_traj, all_cost, all_traj);
}

int f1p_lattice_plan_batch_f32(f1p_ctx* ctx, const double* poses, const double* goals, const double* prev_theta,
                               int32_t E, const f1p_lattice_cfg* cfg, double* steer, double* speed, int32_t* best_idx,
                               double* best_cost, int32_t* status, int32_t* near_idx, float* best_traj32) {
    F1P_ENTER(ctx);
    return lattice_plan_batch_impl<float>(ctx, poses, goals, prev_theta, E, cfg, steer, speed, best_idx, best_cost, status, near_idx, best_traj32, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------------
// One closed-loop control step: poses in, (steer, speed, status) out, nothing else across PCIe.  No copy is submitted in either
// direction: k_lattice_prologue reads the poses straight out of page-locked host memory (one 32-byte read per ego-wave, hidden by
// the other waves) and leaves a device copy for the kernels behind it; the selection kernel stores the three result columns straight
// into page-locked host memory.  The caller's own arrays are used when they are page-locked (f1p_host_alloc / hipHostRegister), the
// context's block otherwise (one memcpy each way on the host).  Previous headings (similarity term) and -- on request -- the winners'
// rows stay in HBM.
static int ensure_step(f1p_ctx* ctx, int E, int S, bool keep_traj) {
    const size_t hb = al256(32 * (size_t)E) + 2 * al256(8 * (size_t)E) + al256(4 * (size_t)E);
    if (hb > ctx->step_host_bytes) {
        F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_step) (void)hipHostFree(ctx->h_step);
        ctx->h_step = nullptr; ctx->step_host_bytes = 0; ctx->h_step_dev = nullptr;
        F1P_HIP(ctx, hipHostMalloc((void**)&ctx->h_step, hb, hipHostMallocDefault));
        ctx->step_host_bytes = hb;
        void* dev = nullptr;
        ctx->h_step_dev = (hipHostGetDevicePointer(&dev, ctx->h_step, 0) == hipSuccess) ? (char*)dev : nullptr;
        if (!ctx->h_step_dev) (void)hipGetLastError();
    }
    const size_t db = al256(32 * (size_t)E) + 2 * al256(4 * (size_t)E) + (keep_traj ? al256(32 * (size_t)E * S) : 0);
    if (db > ctx->step_dev_bytes) {
        F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_step) (void)hipFree(ctx->d_step);
        ctx->d_step = nullptr; ctx->step_dev_bytes = 0; ctx->step_traj_E = 0;
        F1P_HIP(ctx, hipMalloc((void**)&ctx->d_step, db));
        ctx->step_dev_bytes = db;
    }
    return F1P_OK;
}

int f1p_lattice_step_batch(f1p_ctx* ctx, const double* poses, int32_t E, const f1p_lattice_cfg* cfg, double* steer, double* speed,
                           int32_t* status, int32_t keep_traj) {
    F1P_ENTER(ctx);
    int rc = validate_lattice(ctx, cfg, E, true, E == 0 || (poses && steer && speed));
    if (rc) return rc;
    if (cfg->cand_count > 0) return set_error(ctx, F1P_EINVAL, "f1p_lattice_step_batch plans whole egos: cfg.cand_count must be 0");
    if (E == 0) return F1P_OK;
    const int S = cfg->n_stations;
    const size_t e = (size_t)E;
    if ((rc = ensure_step(ctx, E, S, keep_traj != 0))) return rc;
    char* hb = ctx->h_step;
    double* h_poses = (double*)hb; double* h_steer = (double*)(hb + al256(32 * e)); double* h_speed = (double*)((char*)h_steer + al256(8 * e));
    int32_t* h_status = (int32_t*)((char*)h_speed + al256(8 * e));
    char* db = ctx->d_step;
    double* d_pose_copy = (double*)db; int32_t* d_idx = (int32_t*)(db + al256(32 * e)); int32_t* d_near = (int32_t*)((char*)d_idx + al256(4 * e));
    double* d_traj = keep_traj ? (double*)((char*)d_near + al256(4 * e)) : nullptr;
    // the caller's arrays when page-locked, the context's block otherwise
    const double* k_poses = (const double*)pinned_device_ptr(ctx, poses, 32 * e);
    if (!k_poses) { memcpy(h_poses, poses, 32 * e); k_poses = (const double*)pinned_device_ptr(ctx, h_poses, 32 * e); }
    double* k_steer = (double*)pinned_device_ptr(ctx, steer, 8 * e); double* k_speed = (double*)pinned_device_ptr(ctx, speed, 8 * e);
    int32_t* k_status = status ? (int32_t*)pinned_device_ptr(ctx, status, 4 * e) : nullptr;
    const bool own_steer = !k_steer, own_speed = !k_speed, own_status = status && !k_status;
    if (own_steer) k_steer = (double*)pinned_device_ptr(ctx, h_steer, 8 * e);
    if (own_speed) k_speed = (double*)pinned_device_ptr(ctx, h_speed, 8 * e);
    if (own_status) k_status = (int32_t*)pinned_device_ptr(ctx, h_status, 4 * e);
    if (!k_poses || !k_steer || !k_speed || (status && !k_status)) return set_error(ctx, F1P_ESTATE, "page-locked step block is not device-visible");
    // a step IS a link of a closed loop -- for the duration of the step.  The mode the caller set with f1p_lattice_set_closed_loop is restored
    // afterwards (the step used to arm it for good, and a later f1p_lattice_plan_* with prev_theta == NULL and the same batch shape
    // silently picked up -- and overwrote -- the step chain's headings).  The chain's headings stay in the context between steps.
    const bool was_armed = ctx->lattice_closed_loop;
    if (!was_armed && !ctx->step_chain) ctx->cl_valid = false;      // the first step of a chain of its own
    ctx->lattice_closed_loop = true;
    ClosedLoop cl;
    rc = cl_begin(ctx, nullptr, E, S, true, &cl);
    if (rc == F1P_OK)
        rc = launch_lattice(ctx, LATTICE_FULL, k_poses, nullptr, cl.prev, E, cfg, nullptr, nullptr, k_steer, k_speed, d_idx, nullptr, k_status, d_near,
                            d_traj, nullptr, nullptr, nullptr, cl.out, d_pose_copy);
    ctx->lattice_closed_loop = was_armed;
    if (rc) return rc;
    cl_commit(ctx, cl, E, S);
    ctx->step_chain = !was_armed;
    ctx->step_traj_E = keep_traj ? E : 0; ctx->step_traj_S = S;
    F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (own_steer) memcpy(steer, h_steer, 8 * e);
    if (own_speed) memcpy(speed, h_speed, 8 * e);
    if (own_status) memcpy(status, h_status, 4 * e);
    return F1P_OK;
}

int f1p_lattice_fetch_traj(f1p_ctx* ctx, double* best_traj, int32_t E, int32_t S) {
    F1P_ENTER(ctx);
    if (!best_traj) return set_error(ctx, F1P_EINVAL, "best_traj is NULL");
    if (ctx->step_traj_E <= 0 || !ctx->d_step) return set_error(ctx, F1P_ESTATE, "no trajectories kept: call f1p_lattice_step_batch with keep_traj = 1 first");
    if (E != ctx->step_traj_E || S != ctx->step_traj_S) return set_error(ctx, F1P_EINVAL, "E / S differ from the last f1p_lattice_step_batch");
    const size_t e = (size_t)E;
    const char* d_traj = ctx->d_step + al256(32 * e) + 2 * al256(4 * e);
    F1P_HIP(ctx, hipMemcpyAsync(best_traj, d_traj, 32 * e * S, hipMemcpyDeviceToHost, ctx->stream));
    F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F1P_OK;
}

int f1p_lattice_set_mode(f1p_ctx* ctx, int32_t mixed, float* d_cost32, int32_t* d_state) {
    if (!ctx) return F1P_EINVAL;
    if (mixed < 0 || mixed > 3) return set_error(ctx, F1P_EINVAL, "mixed must be 0 (all fp64), 1 (f32 filter, the default), 2 (f32 filter at any batch size) or 3 (as 2, one ego per wave in the per-ego kernels)");
    ctx->lattice_mixed = mixed;
    ctx->d_dbg_lat_cost32 = d_cost32;
    ctx->d_dbg_lat_state = d_state;
    return F1P_OK;
}

int f1p_pure_pursuit_set_form(f1p_ctx* ctx, int32_t egos_per_wave) {
    if (!ctx) return F1P_EINVAL;
    if (egos_per_wave != 0 && egos_per_wave != 1 && egos_per_wave != 4 && egos_per_wave != 8 && egos_per_wave != 16)
        return set_error(ctx, F1P_EINVAL, "egos per wave must be 0 (by batch size), 1, 4, 8 or 16");
    ctx->pursuit_form = egos_per_wave;
    return F1P_OK;
}

int f1p_lattice_set_split(f1p_ctx* ctx, int32_t groups) {
    if (!ctx) return F1P_EINVAL;
    if (groups < 0 || groups > 16) return set_error(ctx, F1P_EINVAL, "groups must be in [0, 16]");
    ctx->lattice_split = groups;
    return F1P_OK;
}

int f1p_lattice_set_clearance(f1p_ctx* ctx, int32_t stations_each_side) {
    if (!ctx) return F1P_EINVAL;
    if (stations_each_side < 0 || stations_each_side > 2) return set_error(ctx, F1P_EINVAL, "stations_each_side must be 0, 1 or 2");
    ctx->lattice_clear_r = stations_each_side;
    return F1P_OK;
}

int f1p_lattice_debug_queue(f1p_ctx* ctx, int32_t* entries_per_ego, int32_t E) {
    F1P_ENTER(ctx);
    if (!entries_per_ego || E < 1) return set_error(ctx, F1P_EINVAL, "entries_per_ego is NULL or E < 1");
    if (!ctx->d_mix_scratch || ctx->mix_last_E != E) return set_error(ctx, F1P_ESTATE, "no mixed-schedule plan of this batch size has run");
    F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
    F1P_HIP(ctx, hipMemcpy(entries_per_ego, ctx->d_mix_scratch + ctx->mix_ego_n_off, sizeof(int32_t) * (size_t)E, hipMemcpyDeviceToHost));
    return F1P_OK;
}

int f1p_lattice_debug_bound(f1p_ctx* ctx, float* d_bound) {
    if (!ctx) return F1P_EINVAL;
    ctx->d_dbg_lat_bound = d_bound;
    return F1P_OK;
}

int f1p_lattice_set_order(f1p_ctx* ctx, int32_t heavy_first) {
    if (!ctx) return F1P_EINVAL;
    if (heavy_first < 0 || heavy_first > 1) return set_error(ctx, F1P_EINVAL, "heavy_first must be 0 or 1");
    ctx->lattice_order = heavy_first;
    return F1P_OK;
}

int f1p_lattice_debug_pass(f1p_ctx* ctx, int32_t* d_pass) {
    if (!ctx) return F1P_EINVAL;
    ctx->d_dbg_lat_pass = d_pass;
    return F1P_OK;
}

int f1p_lattice_debug_margins(f1p_ctx* ctx, int32_t enable, float margin_rel, float margin_abs) {
    if (!ctx) return F1P_EINVAL;
    ctx->dbg_margins = enable != 0; ctx->dbg_margin_rel = margin_rel; ctx->dbg_margin_abs = margin_abs;
    return F1P_OK;
}

int f1p_lattice_set_audit(f1p_ctx* ctx, int32_t every_n, int32_t n_egos) {
    if (!ctx) return F1P_EINVAL;
    if (every_n < 0 || n_egos < 0 || (every_n > 0 && n_egos < 1)) return set_error(ctx, F1P_EINVAL, "every_n >= 0 and, when auditing, n_egos >= 1");
    ctx->audit_every = every_n; ctx->audit_egos = n_egos;
    return F1P_OK;
}

int f1p_lattice_audit_read(f1p_ctx* ctx, uint64_t out[3], int32_t reset) {
    F1P_ENTER(ctx);
    if (!out) return set_error(ctx, F1P_EINVAL, "out is NULL");
    out[0] = out[1] = out[2] = 0;
    if (!ctx->d_audit) return F1P_OK;
    F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long h[3];
    F1P_HIP(ctx, hipMemcpy(h, ctx->d_audit, sizeof(h), hipMemcpyDeviceToHost));
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2];
    if (reset) F1P_HIP(ctx, hipMemset(ctx->d_audit, 0, sizeof(h)));
    return F1P_OK;
}

int f1p_lattice_set_pipeline(f1p_ctx* ctx, int32_t chunks) {
    if (!ctx) return F1P_EINVAL;
    if (chunks < 0 || chunks > 8) return set_error(ctx, F1P_EINVAL, "chunks must be in [0, 8]");
    ctx->lattice_chunks = chunks;
    return F1P_OK;
}

int f1p_lattice_profile(f1p_ctx* ctx, int32_t enable, float kernel_ms[4]) {
    F1P_ENTER(ctx);
    if (kernel_ms) {
        if (!ctx->lattice_profile || !ctx->lattice_profile_valid) return set_error(ctx, F1P_ESTATE, "no profiled mixed-schedule plan has run");
        F1P_HIP(ctx, hipEventSynchronize(ctx->ev_prof[4]));
        for (int k = 0; k < 4; ++k) F1P_HIP(ctx, hipEventElapsedTime(&kernel_ms[k], ctx->ev_prof[k], ctx->ev_prof[k + 1]));
    }
    if (enable && !ctx->ev_prof[0])
        for (auto& ev : ctx->ev_prof) F1P_HIP(ctx, hipEventCreate(&ev));
    ctx->lattice_profile = enable != 0;
    if (!enable) ctx->lattice_profile_valid = false;
    return F1P_OK;
}

int f1p_clothoid_sample_batch(f1p_ctx* ctx, const double* params, int32_t n, int32_t npts, double* rows) {
    F1P_ENTER(ctx);
    if (n < 0 || (n > 0 && (!params || !rows))) return set_error(ctx, F1P_EINVAL, "bad params / rows / n");
    if (npts < 1 || npts > 65536) return set_error(ctx, F1P_EINVAL, "npts must be in [1, 65536]");
    Stage s(ctx);
    s.need(8 * 3 * (size_t)n); s.need(8 * 4 * (size_t)n * npts);
    int rc = s.begin(); if (rc) return rc;
    const double* d_p;
    if ((rc = s.in(params, (size_t)3 * n, &d_p))) return rc;
    double* d_rows = s.out(rows, (size_t)4 * n * npts);
    if ((rc = launch_clothoid_sample(ctx, d_p, n, npts, d_rows))) return rc;
    return s.finish();
}

int f1p_clothoid_g1_batch(f1p_ctx* ctx, const double* goals, int32_t n, double* kappa0, double* dkappa, double* length, int32_t* ok) {
    F1P_ENTER(ctx);
    if (n < 0 || (n > 0 && !goals)) return set_error(ctx, F1P_EINVAL, "bad goals / n");
    Stage s(ctx);
    s.need(8 * 3 * (size_t)n); s.need(8 * (size_t)n, kappa0); s.need(8 * (size_t)n, dkappa); s.need(8 * (size_t)n, length); s.need(4 * (size_t)n, ok);
    int rc = s.begin(); if (rc) return rc;
    const double* d_g;
    if ((rc = s.in(goals, (size_t)3 * n, &d_g))) return rc;
    double* d_k0 = s.out(kappa0, n); double* d_dk = s.out(dkappa, n); double* d_L = s.out(length, n); int32_t* d_ok = s.out(ok, n);
    if ((rc = launch_clothoid_g1(ctx, d_g, n, d_k0, d_dk, d_L, d_ok))) return rc;
    return s.finish();
}

// ---------------------------------------------------------------------------------------------------
int f1p_kmpc_shoot_dev(f1p_ctx* ctx, const double* d_x0, const double* d_ref, const float* d_controls, int32_t E,
                       const f1p_kmpc_cfg* cfg, double* d_steer, double* d_speed, int32_t* d_best_idx,
                       double* d_best_cost, double* d_best_seq) {
    F1P_ENTER(ctx);
    int rc = validate_kmpc(ctx, cfg, E); if (rc) return rc;
    if (E > 0 && (!d_x0 || !d_ref || !d_controls || !d_steer || !d_speed || !d_best_idx))
        return set_error(ctx, F1P_EINVAL, "x0, ref, controls, steer, speed and best_idx are required");
    return launch_kmpc_shoot(ctx, d_x0, d_ref, d_controls, E, cfg, d_steer, d_speed, d_best_idx, d_best_cost, d_best_seq);
}

int f1p_kmpc_shoot_batch(f1p_ctx* ctx, const double* x0, const double* ref, const float* controls, int32_t E,
                         const f1p_kmpc_cfg* cfg, double* steer, double* speed, int32_t* best_idx, double* best_cost,
                         double* best_seq) {
    F1P_ENTER(ctx);
    int rc = validate_kmpc(ctx, cfg, E); if (rc) return rc;
    if (E > 0 && (!x0 || !ref || !controls || !steer || !speed || !best_idx))
        return set_error(ctx, F1P_EINVAL, "x0, ref, controls, steer, speed and best_idx are required");
    const size_t T = cfg->horizon, R = cfg->n_rollouts, e = E;
    Stage s(ctx);
    s.need(8 * 4 * e); s.need(8 * e * 4 * (T + 1)); s.need(4 * e * T * 2 * R);
    s.need(8 * e); s.need(8 * e); s.need(4 * e); s.need(8 * e, best_cost); s.need(8 * e * T * 2, best_seq);
    if ((rc = s.begin())) return rc;
    const double *d_x0, *d_ref; const float* d_c;
    if ((rc = s.in(x0, 4 * e, &d_x0))) return rc;
    if ((rc = s.in(ref, e * 4 * (T + 1), &d_ref))) return rc;
    if ((rc = s.in(controls, e * T * 2 * R, &d_c))) return rc;
    double* d_steer = s.out(steer, e); double* d_speed = s.out(speed, e); int32_t* d_bi = s.out(best_idx, e);
    double* d_bc = s.out(best_cost, e); double* d_bs = s.out(best_seq, e * T * 2);
    if ((rc = launch_kmpc_shoot(ctx, d_x0, d_ref, d_c, E, cfg, d_steer, d_speed, d_bi, d_bc, d_bs))) return rc;
    return s.finish();
}

int f1p_kmpc_set_mode(f1p_ctx* ctx, int32_t mixed, float* d_cost32, int32_t* d_n_refined) {
    if (!ctx) return F1P_EINVAL;
    ctx->kmpc_mixed = mixed != 0;
    ctx->d_dbg_cost32 = d_cost32;
    ctx->d_dbg_nref = d_n_refined;
    return F1P_OK;
}

int f1p_kmpc_predict_batch(f1p_ctx* ctx, const double* x0, const double* oa, const double* od, int32_t E,
                           const f1p_kmpc_cfg* cfg, double* path) {
    F1P_ENTER(ctx);
    int rc = validate_kmpc(ctx, cfg, E); if (rc) return rc;
    if (E > 0 && (!x0 || !oa || !od || !path)) return set_error(ctx, F1P_EINVAL, "x0, oa, od and path are required");
    const size_t T = cfg->horizon, e = E;
    Stage s(ctx);
    s.need(8 * 4 * e); s.need(8 * e * T); s.need(8 * e * T); s.need(8 * e * 4 * (T + 1));
    if ((rc = s.begin())) return rc;
    const double *d_x0, *d_oa, *d_od;
    if ((rc = s.in(x0, 4 * e, &d_x0))) return rc;
    if ((rc = s.in(oa, e * T, &d_oa))) return rc;
    if ((rc = s.in(od, e * T, &d_od))) return rc;
    double* d_path = s.out(path, e * 4 * (T + 1));
    if ((rc = launch_kmpc_predict(ctx, d_x0, d_oa, d_od, E, cfg, d_path))) return rc;
    return s.finish();
}

int f1p_kmpc_ref_batch(f1p_ctx* ctx, const double* states, int32_t E, int32_t horizon, double dt, double dl, double* ref) {
    F1P_ENTER(ctx);
    if (E < 0 || (E > 0 && (!states || !ref))) return set_error(ctx, F1P_EINVAL, "bad states / ref / E");
    if (horizon < 1 || !(dt > 0) || !(dl > 0)) return set_error(ctx, F1P_EINVAL, "horizon, dt and dl must be positive");
    if (ctx->n_wp < 2 || !ctx->has_psi) return set_error(ctx, F1P_ESTATE, "waypoints with a heading column are required");
    Stage s(ctx);
    s.need(8 * 4 * (size_t)E); s.need(8 * (size_t)E * 4 * (horizon + 1));
    int rc = s.begin(); if (rc) return rc;
    const double* d_s;
    if ((rc = s.in(states, (size_t)4 * E, &d_s))) return rc;
    double* d_ref = s.out(ref, (size_t)E * 4 * (horizon + 1));
    if ((rc = launch_kmpc_ref(ctx, d_s, E, horizon, dt, dl, d_ref))) return rc;
    return s.finish();
}

int f1p_kmpc_sample_controls_dev(f1p_ctx* ctx, float* d_controls, int32_t E, const f1p_kmpc_cfg* cfg, uint64_t seed,
                                 double sigma_accel, double sigma_steer) {
    F1P_ENTER(ctx);
    int rc = validate_kmpc(ctx, cfg, E); if (rc) return rc;
    if (E > 0 && !d_controls) return set_error(ctx, F1P_EINVAL, "controls is NULL");
    return launch_kmpc_sample(ctx, d_controls, E, cfg, seed, sigma_accel, sigma_steer);
}

static std::string rccl_err(f1p_ctx* ctx, const char* what, ncclResult_t r) {
    auto es = rccl_sym<pfn_ncclGetErrorString>(ctx, "ncclGetErrorString");
    return std::string(what) + " failed: " + (es ? es(r) : "?") + " (code " + std::to_string((int)r) + ")";
}

// ---------------------------------------------------------------------------------------------------
// shooting MPC with in-kernel control generation and a device-resident warm start
// ---------------------------------------------------------------------------------------------------
static int validate_sampler(f1p_ctx* ctx, const f1p_kmpc_sampler* smp) {
    if (!smp) return set_error(ctx, F1P_EINVAL, "sampler is NULL");
    if (!(smp->sigma_accel >= 0.0) || !(smp->sigma_steer >= 0.0) || !isfinite(smp->sigma_accel) || !isfinite(smp->sigma_steer))
        return set_error(ctx, F1P_EINVAL, "sampler sigmas must be finite and >= 0");
    return F1P_OK;
}

// the ctx's warm buffer for (E, T); a change of shape drops the old contents
static int ensure_warm(f1p_ctx* ctx, int E, int T) {
    if (ctx->d_kmpc_warm && ctx->kmpc_warm_E == E && ctx->kmpc_warm_T == T) return F1P_OK;
    F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_kmpc_warm) (void)hipFree(ctx->d_kmpc_warm);
    ctx->d_kmpc_warm = nullptr; ctx->kmpc_warm_valid = false; ctx->kmpc_warm_E = ctx->kmpc_warm_T = 0;
    F1P_HIP(ctx, hipMalloc((void**)&ctx->d_kmpc_warm, sizeof(float) * 2 * (size_t)E * T));
    ctx->kmpc_warm_E = E; ctx->kmpc_warm_T = T;
    return F1P_OK;
}

int f1p_kmpc_plan_dev(f1p_ctx* ctx, const double* d_x0, const double* d_ref, int32_t E, const f1p_kmpc_cfg* cfg,
                      const f1p_kmpc_sampler* smp, double* d_steer, double* d_speed, int32_t* d_best_idx,
                      double* d_best_cost, double* d_best_seq) {
    F1P_ENTER(ctx);
    int rc = validate_kmpc(ctx, cfg, E); if (rc) return rc;
    if ((rc = validate_sampler(ctx, smp))) return rc;
    if (E == 0) return F1P_OK;
    if (!d_x0 || !d_ref || !d_steer || !d_speed || !d_best_idx) return set_error(ctx, F1P_EINVAL, "x0, ref, steer, speed and best_idx are required");
    if (cfg->n_rollouts > 8192) return set_error(ctx, F1P_EINVAL, "at most 8192 rollouts per plan");
    if ((rc = ensure_warm(ctx, E, cfg->horizon))) return rc;
    const float* warm_in = (smp->use_warm && ctx->kmpc_warm_valid) ? ctx->d_kmpc_warm : nullptr;
    rc = launch_kmpc_plan_gen(ctx, d_x0, d_ref, E, cfg, smp, warm_in, ctx->d_kmpc_warm, d_steer, d_speed, d_best_idx, d_best_cost, d_best_seq);
    if (rc == F1P_OK) ctx->kmpc_warm_valid = true;
    return rc;
}

int f1p_kmpc_plan_batch(f1p_ctx* ctx, const double* x0, int32_t E, const f1p_kmpc_cfg* cfg, double dl,
                        const f1p_kmpc_sampler* smp, double* steer, double* speed, int32_t* best_idx, double* best_cost,
                        double* best_seq) {
    F1P_ENTER(ctx);
    int rc = validate_kmpc(ctx, cfg, E); if (rc) return rc;
    if ((rc = validate_sampler(ctx, smp))) return rc;
    if (E > 0 && (!x0 || !steer || !speed || !best_idx)) return set_error(ctx, F1P_EINVAL, "x0, steer, speed and best_idx are required");
    if (!(dl > 0)) return set_error(ctx, F1P_EINVAL, "dl must be > 0");
    if (ctx->n_wp < 2 || !ctx->has_psi) return set_error(ctx, F1P_ESTATE, "waypoints with a heading column are required");
    const size_t T = cfg->horizon, e = E;
    Stage s(ctx);
    s.need(8 * 4 * e); s.need(8 * e * 4 * (T + 1));
    s.need(8 * e); s.need(8 * e); s.need(4 * e); s.need(8 * e, best_cost); s.need(8 * e * T * 2, best_seq);
    if ((rc = s.begin())) return rc;
    const double* d_x0;
    if ((rc = s.in(x0, 4 * e, &d_x0))) return rc;
    double* d_ref = (double*)arena_take(ctx, 8 * e * 4 * (T + 1));
    double* d_steer = s.out(steer, e); double* d_speed = s.out(speed, e); int32_t* d_bi = s.out(best_idx, e);
    double* d_bc = s.out(best_cost, e); double* d_bs = s.out(best_seq, e * T * 2);
    if ((rc = launch_kmpc_ref(ctx, d_x0, E, cfg->horizon, cfg->dt, dl, d_ref))) return rc;            // calc_ref_trajectory_kinematic :162-206
    if ((rc = f1p_kmpc_plan_dev(ctx, d_x0, d_ref, E, cfg, smp, d_steer, d_speed, d_bi, d_bc, d_bs))) return rc;
    return s.finish();
}

int f1p_kmpc_gen_controls_dev(f1p_ctx* ctx, float* d_controls, int32_t E, const f1p_kmpc_cfg* cfg, const f1p_kmpc_sampler* smp) {
    F1P_ENTER(ctx);
    int rc = validate_kmpc(ctx, cfg, E); if (rc) return rc;
    if ((rc = validate_sampler(ctx, smp))) return rc;
    if (E > 0 && !d_controls) return set_error(ctx, F1P_EINVAL, "controls is NULL");
    const bool warm = smp->use_warm && ctx->kmpc_warm_valid && ctx->kmpc_warm_E == E && ctx->kmpc_warm_T == cfg->horizon;
    return launch_kmpc_gen_controls(ctx, d_controls, E, cfg, smp, warm ? ctx->d_kmpc_warm : nullptr);
}

int f1p_kmpc_warm_reset(f1p_ctx* ctx) {
    if (!ctx) return F1P_EINVAL;
    ctx->kmpc_warm_valid = false;
    return F1P_OK;
}

int f1p_kmpc_warm_get(f1p_ctx* ctx, float* warm, int32_t E, int32_t T) {
    F1P_ENTER(ctx);
    if (!warm) return set_error(ctx, F1P_EINVAL, "warm is NULL");
    if (!ctx->kmpc_warm_valid || ctx->kmpc_warm_E != E || ctx->kmpc_warm_T != T) return set_error(ctx, F1P_ESTATE, "no warm start of this shape is held");
    F1P_HIP(ctx, hipMemcpyAsync(warm, ctx->d_kmpc_warm, sizeof(float) * 2 * (size_t)E * T, hipMemcpyDeviceToHost, ctx->stream));
    F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F1P_OK;
}

int f1p_kmpc_warm_set(f1p_ctx* ctx, const float* warm, int32_t E, int32_t T) {
    F1P_ENTER(ctx);
    if (!warm || E < 1 || T < 1) return set_error(ctx, F1P_EINVAL, "bad warm / E / T");
    int rc = ensure_warm(ctx, E, T); if (rc) return rc;
    F1P_HIP(ctx, hipMemcpyAsync(ctx->d_kmpc_warm, warm, sizeof(float) * 2 * (size_t)E * T, hipMemcpyHostToDevice, ctx->stream));
    F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->kmpc_warm_valid = true;
    return F1P_OK;
}

int f1p_kmpc_set_yaw_fixup(f1p_ctx* ctx, int32_t on) {
    F1P_ENTER(ctx);
    ctx->kmpc_yaw_fixup = on ? 1 : 0;
    return F1P_OK;
}

int f1p_kmpc_set_groups(f1p_ctx* ctx, int32_t groups) {
    if (!ctx) return F1P_EINVAL;
    if (groups < 0 || groups > 64) return set_error(ctx, F1P_EINVAL, "groups must be in [0, 64]");
    ctx->kmpc_groups = groups;
    return F1P_OK;
}

int f1p_comm_unique_id(f1p_ctx* ctx, uint8_t id[F1P_COMM_ID_BYTES]) {
    F1P_ENTER(ctx);
    int rc = rccl_open(ctx); if (rc) return rc;
    auto fn = rccl_sym<pfn_ncclGetUniqueId>(ctx, "ncclGetUniqueId");
    if (!fn) return set_error(ctx, F1P_ECOMM, "ncclGetUniqueId not found");
    ncclUniqueId uid;
    const ncclResult_t r = fn(&uid);
    if (r != ncclSuccess) return set_error(ctx, F1P_ECOMM, rccl_err(ctx, "ncclGetUniqueId", r));
    memcpy(id, uid.internal, F1P_COMM_ID_BYTES);
    return F1P_OK;
}

int f1p_comm_init(f1p_ctx* ctx, const uint8_t id[F1P_COMM_ID_BYTES], int32_t nranks, int32_t rank) {
    F1P_ENTER(ctx);
    if (nranks < 1 || rank < 0 || rank >= nranks) return set_error(ctx, F1P_EINVAL, "bad nranks / rank");
    int rc = rccl_open(ctx); if (rc) return rc;
    if (ctx->comm) f1p_comm_destroy(ctx);
    auto fn = rccl_sym<pfn_ncclCommInitRank>(ctx, "ncclCommInitRank");
    if (!fn) return set_error(ctx, F1P_ECOMM, "ncclCommInitRank not found");
    ncclUniqueId uid;
    memcpy(uid.internal, id, F1P_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = fn(&comm, nranks, uid, rank);
    if (r != ncclSuccess) { ctx->comm = nullptr; return set_error(ctx, F1P_ECOMM, rccl_err(ctx, "ncclCommInitRank", r)); }
    ctx->comm = comm;
    ctx->comm_rank = rank; ctx->comm_nranks = nranks;
    return F1P_OK;
}

int f1p_comm_info(f1p_ctx* ctx, int32_t* nranks, int32_t* rank) {
    F1P_ENTER(ctx);
    if (!ctx->comm) return set_error(ctx, F1P_ESTATE, "communicator not initialised: call f1p_comm_init");
    auto cnt = rccl_sym<pfn_ncclCommCount>(ctx, "ncclCommCount");
    auto ur = rccl_sym<pfn_ncclCommUserRank>(ctx, "ncclCommUserRank");
    if (!cnt || !ur) return set_error(ctx, F1P_ECOMM, "ncclCommCount / ncclCommUserRank not found");
    int n = 0, r = 0;
    ncclResult_t e = cnt((ncclComm_t)ctx->comm, &n);
    if (e == ncclSuccess) e = ur((ncclComm_t)ctx->comm, &r);
    if (e != ncclSuccess) return set_error(ctx, F1P_ECOMM, rccl_err(ctx, "ncclCommCount", e));
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    return F1P_OK;
}

int f1p_comm_destroy(f1p_ctx* ctx) {
    if (!ctx) return F1P_EINVAL;
    if (ctx->comm && ctx->rccl_lib) {
        auto fn = rccl_sym<pfn_ncclCommDestroy>(ctx, "ncclCommDestroy");
        if (fn) (void)fn((ncclComm_t)ctx->comm);
    }
    ctx->comm = nullptr;
    return F1P_OK;
}

// Cross-rank argmin with np.argmin's rules (first minimum; a NaN cost is "smaller" than any number, lattice_planner.py:159-172,
// f1p::argmin_better).  ncclMin on floating point leaves NaN handling unspecified, so the cost travels as a monotone
// unsigned 64-bit key (k_argmin_key: NaN -> 0, otherwise the IEEE bits made order-preserving) and both reductions are
// integer minima: all-reduce(min, u64) on the key, then all-reduce(min, i32) on the index among the ranks holding that key.
// The key map is a bijection on non-NaN doubles (-0.0 is folded into +0.0, which np.argmin also treats as equal), so the
// cost that comes back is bit-identical to the single-GPU result.
int f1p_comm_argmin_dev(f1p_ctx* ctx, double* d_cost, int32_t* d_idx, int32_t E) {
    F1P_ENTER(ctx);
    if (!ctx->comm) return set_error(ctx, F1P_ESTATE, "communicator not initialised: call f1p_comm_init");
    if (E < 0 || (E > 0 && (!d_cost || !d_idx))) return set_error(ctx, F1P_EINVAL, "bad cost / idx / E");
    if (E == 0) return F1P_OK;
    if (E > ctx->comm_cap) {
        F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_comm_key) (void)hipFree(ctx->d_comm_key);
        if (ctx->d_comm_idx) (void)hipFree(ctx->d_comm_idx);
        ctx->d_comm_key = nullptr; ctx->d_comm_idx = nullptr; ctx->comm_cap = 0;
        F1P_HIP(ctx, hipMalloc((void**)&ctx->d_comm_key, sizeof(uint64_t) * 2 * (size_t)E));   // [own keys | reduced keys]
        F1P_HIP(ctx, hipMalloc((void**)&ctx->d_comm_idx, sizeof(int32_t) * (size_t)E));
        ctx->comm_cap = E;
    }
    if (ctx->comm_exchange == 1) {
        // ONE collective: all-gather of the (key, index) records (16 B per ego and rank), then every rank takes the minimum itself.  Half
        // the xGMI latency of the two dependent all-reduces below at the price of N x 16 B instead of 12 B per ego on the wire.
        const int N = ctx->comm_nranks;
        if (E > ctx->comm_rec_cap || N != ctx->comm_rec_ranks) {
            F1P_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_comm_rec) (void)hipFree(ctx->d_comm_rec);
            ctx->d_comm_rec = nullptr; ctx->comm_rec_cap = 0;
            F1P_HIP(ctx, hipMalloc((void**)&ctx->d_comm_rec, sizeof(uint64_t) * 2 * (size_t)E * (size_t)(N + 1)));
            ctx->comm_rec_cap = E; ctx->comm_rec_ranks = N;
        }
        auto ag = rccl_sym<pfn_ncclAllGather>(ctx, "ncclAllGather");
        if (!ag) return set_error(ctx, F1P_ECOMM, "ncclAllGather not found");
        uint64_t* mine = ctx->d_comm_rec;
        uint64_t* all = ctx->d_comm_rec + 2 * (size_t)ctx->comm_rec_cap;
        int rc1 = launch_argmin_pack(ctx, d_cost, d_idx, mine, E); if (rc1) return rc1;
        ncclResult_t r1 = ag(mine, all, 2 * (size_t)E, ncclUint64, (ncclComm_t)ctx->comm, ctx->stream);
        if (r1 != ncclSuccess) return set_error(ctx, F1P_ECOMM, rccl_err(ctx, "ncclAllGather(u64)", r1));
        return launch_argmin_reduce(ctx, all, N, E, d_idx, d_cost);
    }
    auto ar = rccl_sym<pfn_ncclAllReduce>(ctx, "ncclAllReduce");
    if (!ar) return set_error(ctx, F1P_ECOMM, "ncclAllReduce not found");
    uint64_t* own = ctx->d_comm_key;
    uint64_t* red = ctx->d_comm_key + E;
    int rc = launch_argmin_key(ctx, d_cost, own, E); if (rc) return rc;
    // 1. global minimum key per ego
    ncclResult_t r = ar(own, red, (size_t)E, ncclUint64, ncclMin, (ncclComm_t)ctx->comm, ctx->stream);
    if (r != ncclSuccess) return set_error(ctx, F1P_ECOMM, rccl_err(ctx, "ncclAllReduce(min, u64)", r));
    // 2. ranks that hold that key keep their index, the others contribute INT32_MAX; the winning cost is decoded in place
    rc = launch_argmin_mask(ctx, own, red, d_idx, ctx->d_comm_idx, d_cost, E); if (rc) return rc;
    // 3. lowest index among the holders (np.argmin first-minimum rule)
    r = ar(ctx->d_comm_idx, d_idx, (size_t)E, ncclInt32, ncclMin, (ncclComm_t)ctx->comm, ctx->stream);
    if (r != ncclSuccess) return set_error(ctx, F1P_ECOMM, rccl_err(ctx, "ncclAllReduce(min, i32)", r));
    return F1P_OK;
}

int f1p_comm_set_exchange(f1p_ctx* ctx, int32_t mode) {
    if (!ctx) return F1P_EINVAL;
    if (mode != 0 && mode != 1) return set_error(ctx, F1P_EINVAL, "mode must be 0 (two all-reduces) or 1 (one all-gather + local minimum)");
    ctx->comm_exchange = mode;
    return F1P_OK;
}

// the local kernels of the single-collective exchange on host arrays (the all-gather replaced by the caller): cost / idx [N][E] of N
// emulated ranks -> idx_out [E], cost_out [E]
int f1p_argmin_gather_reduce_batch(f1p_ctx* ctx, const double* cost, const int32_t* idx, int32_t N, int32_t E, int32_t* idx_out, double* cost_out) {
    F1P_ENTER(ctx);
    if (N < 1 || E < 0 || (E > 0 && (!cost || !idx || !idx_out || !cost_out))) return set_error(ctx, F1P_EINVAL, "bad argument");
    const size_t n = (size_t)N * E;
    Stage s(ctx);
    s.need(8 * n); s.need(4 * n); s.need(16 * n); s.need(4 * (size_t)E); s.need(8 * (size_t)E);
    int rc = s.begin(); if (rc) return rc;
    const double* d_c; const int32_t* d_i;
    if ((rc = s.in(cost, n, &d_c))) return rc;
    if ((rc = s.in(idx, n, &d_i))) return rc;
    uint64_t* d_rec = (uint64_t*)arena_take(ctx, 16 * n);
    int32_t* d_io = s.out(idx_out, (size_t)E); double* d_co = s.out(cost_out, (size_t)E);
    for (int r = 0; r < N; ++r)
        if ((rc = launch_argmin_pack(ctx, d_c + (size_t)r * E, d_i + (size_t)r * E, d_rec + 2 * (size_t)r * E, E))) return rc;
    if ((rc = launch_argmin_reduce(ctx, d_rec, N, E, d_io, d_co))) return rc;
    return s.finish();
}

// the two local kernels of the exchange on host arrays (the collective replaced by the caller): lets a single-GPU box check
// the key map against np.argmin for NaN / inf / signed-zero costs.  keys_out [E] <- key(cost[e]);
int f1p_argmin_key_batch(f1p_ctx* ctx, const double* cost, int32_t E, uint64_t* keys_out) {
    F1P_ENTER(ctx);
    if (E < 0 || (E > 0 && (!cost || !keys_out))) return set_error(ctx, F1P_EINVAL, "bad cost / keys / E");
    Stage s(ctx);
    s.need(8 * (size_t)E); s.need(8 * (size_t)E);
    int rc = s.begin(); if (rc) return rc;
    const double* d_c;
    if ((rc = s.in(cost, (size_t)E, &d_c))) return rc;
    uint64_t* d_k = s.out(keys_out, (size_t)E);
    if ((rc = launch_argmin_key(ctx, d_c, d_k, E))) return rc;
    return s.finish();
}
// own_keys / min_keys / idx [E] -> masked_idx [E] (idx where own == min, else INT32_MAX), cost_out [E] = decoded min key
int f1p_argmin_mask_batch(f1p_ctx* ctx, const uint64_t* own_keys, const uint64_t* min_keys, const int32_t* idx, int32_t E,
                          int32_t* masked_idx, double* cost_out) {
    F1P_ENTER(ctx);
    if (E < 0 || (E > 0 && (!own_keys || !min_keys || !idx || !masked_idx || !cost_out))) return set_error(ctx, F1P_EINVAL, "NULL argument");
    Stage s(ctx);
    s.need(8 * (size_t)E); s.need(8 * (size_t)E); s.need(4 * (size_t)E); s.need(4 * (size_t)E); s.need(8 * (size_t)E);
    int rc = s.begin(); if (rc) return rc;
    const uint64_t *d_o, *d_m; const int32_t* d_i;
    if ((rc = s.in(own_keys, (size_t)E, &d_o))) return rc;
    if ((rc = s.in(min_keys, (size_t)E, &d_m))) return rc;
    if ((rc = s.in(idx, (size_t)E, &d_i))) return rc;
    int32_t* d_mi = s.out(masked_idx, (size_t)E); double* d_c = s.out(cost_out, (size_t)E);
    if ((rc = launch_argmin_mask(ctx, d_o, d_m, d_i, d_mi, d_c, E))) return rc;
    return s.finish();
}

// ---------------------------------------------------------------------------------------------------
// dynamic single-track shooting
// ---------------------------------------------------------------------------------------------------
void f1p_stmpc_cfg_default(f1p_stmpc_cfg* cfg) {
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->horizon = 40; cfg->n_rollouts = 512;
    cfg->dt = 0.025; cfg->wheelbase = 0.33; cfg->max_steer = 0.4189; cfg->max_steer_v = 3.2;
    cfg->max_speed = 6.0; cfg->min_speed = 0.0; cfg->max_accel = 3.0;
    const double q[7] = {32.0, 32.0, 0.0, 1.0, 0.5, 0.0, 0.0};
    for (int i = 0; i < 7; ++i) { cfg->q[i] = q[i]; cfg->qf[i] = q[i]; }
    cfg->r[0] = 0.5; cfg->r[1] = 0.01; cfg->rd[0] = 0.3; cfg->rd[1] = 0.01;
    const double p[8] = {3.74, 0.15875, 0.17145, 0.074, 4.718, 5.4562, 0.04712, 1.0489};
    for (int i = 0; i < 8; ++i) cfg->params[i] = p[i];
}

static int validate_stmpc(f1p_ctx* ctx, const f1p_stmpc_cfg* cfg, int E) {
    if (!cfg) return set_error(ctx, F1P_EINVAL, "cfg is NULL");
    if (E < 0) return set_error(ctx, F1P_EINVAL, "E must be >= 0");
    if (cfg->horizon < 1 || cfg->horizon > 4096) return set_error(ctx, F1P_EINVAL, "horizon must be in [1, 4096]");
    if (cfg->n_rollouts < 1) return set_error(ctx, F1P_EINVAL, "n_rollouts must be >= 1");
    if (!(cfg->dt > 0) || !(cfg->wheelbase > 0)) return set_error(ctx, F1P_EINVAL, "dt and wheelbase must be > 0");
    return F1P_OK;
}

int f1p_stmpc_predict_batch(f1p_ctx* ctx, const double* x0, const double* oa, const double* od_v, int32_t E,
                            const f1p_stmpc_cfg* cfg, double* path) {
    F1P_ENTER(ctx);
    int rc = validate_stmpc(ctx, cfg, E); if (rc) return rc;
    if (E > 0 && (!x0 || !oa || !od_v || !path)) return set_error(ctx, F1P_EINVAL, "x0, oa, od_v and path are required");
    const size_t T = cfg->horizon, e = E;
    Stage s(ctx);
    s.need(8 * 7 * e); s.need(8 * e * T); s.need(8 * e * T); s.need(8 * e * 7 * (T + 1));
    if ((rc = s.begin())) return rc;
    const double *d_x0, *d_oa, *d_od;
    if ((rc = s.in(x0, 7 * e, &d_x0))) return rc;
    if ((rc = s.in(oa, e * T, &d_oa))) return rc;
    if ((rc = s.in(od_v, e * T, &d_od))) return rc;
    double* d_path = s.out(path, e * 7 * (T + 1));
    if ((rc = launch_stmpc_predict(ctx, d_x0, d_oa, d_od, E, cfg, d_path))) return rc;
    return s.finish();
}

int f1p_stmpc_ref_batch(f1p_ctx* ctx, const double* states, int32_t E, int32_t horizon, double dt, double dl, double* ref) {
    F1P_ENTER(ctx);
    if (E < 0 || (E > 0 && (!states || !ref))) return set_error(ctx, F1P_EINVAL, "bad states / ref / E");
    if (horizon < 1 || !(dt > 0) || !(dl > 0)) return set_error(ctx, F1P_EINVAL, "horizon, dt and dl must be positive");
    if (ctx->n_wp < 2 || !ctx->has_psi) return set_error(ctx, F1P_ESTATE, "waypoints with a heading column are required");
    Stage s(ctx);
    s.need(8 * 4 * (size_t)E); s.need(8 * (size_t)E * 7 * (horizon + 1));
    int rc = s.begin(); if (rc) return rc;
    const double* d_s;
    if ((rc = s.in(states, (size_t)4 * E, &d_s))) return rc;
    double* d_ref = s.out(ref, (size_t)E * 7 * (horizon + 1));
    if ((rc = launch_stmpc_ref(ctx, d_s, E, horizon, dt, dl, d_ref))) return rc;
    return s.finish();
}

int f1p_stmpc_set_mode(f1p_ctx* ctx, int32_t mixed, float* d_cost32, int32_t* d_n_refined) {
    if (!ctx) return F1P_EINVAL;
    ctx->stmpc_mixed = mixed != 0;
    ctx->d_dbg_st_cost32 = d_cost32; ctx->d_dbg_st_nref = d_n_refined;
    return F1P_OK;
}

int f1p_stmpc_shoot_dev(f1p_ctx* ctx, const double* d_x0, const double* d_ref, const float* d_controls, int32_t E,
                        const f1p_stmpc_cfg* cfg, double* d_steer, double* d_speed, int32_t* d_best_idx,
                        double* d_best_cost, double* d_best_seq) {
    F1P_ENTER(ctx);
    int rc = validate_stmpc(ctx, cfg, E); if (rc) return rc;
    if (E > 0 && (!d_x0 || !d_ref || !d_controls || !d_steer || !d_speed || !d_best_idx))
        return set_error(ctx, F1P_EINVAL, "x0, ref, controls, steer, speed and best_idx are required");
    return launch_stmpc_shoot(ctx, d_x0, d_ref, d_controls, E, cfg, d_steer, d_speed, d_best_idx, d_best_cost, d_best_seq);
}

int f1p_stmpc_shoot_batch(f1p_ctx* ctx, const double* x0, const double* ref, const float* controls, int32_t E,
                          const f1p_stmpc_cfg* cfg, double* steer, double* speed, int32_t* best_idx, double* best_cost,
                          double* best_seq) {
    F1P_ENTER(ctx);
    int rc = validate_stmpc(ctx, cfg, E); if (rc) return rc;
    if (E > 0 && (!x0 || !ref || !controls || !steer || !speed || !best_idx))
        return set_error(ctx, F1P_EINVAL, "x0, ref, controls, steer, speed and best_idx are required");
    const size_t T = cfg->horizon, R = cfg->n_rollouts, e = E;
    Stage s(ctx);
    s.need(8 * 7 * e); s.need(8 * e * 7 * (T + 1)); s.need(4 * e * T * 2 * R);
    s.need(8 * e); s.need(8 * e); s.need(4 * e); s.need(8 * e, best_cost); s.need(8 * e * T * 2, best_seq);
    if ((rc = s.begin())) return rc;
    const double *d_x0, *d_ref; const float* d_c;
    if ((rc = s.in(x0, 7 * e, &d_x0))) return rc;
    if ((rc = s.in(ref, e * 7 * (T + 1), &d_ref))) return rc;
    if ((rc = s.in(controls, e * T * 2 * R, &d_c))) return rc;
    double* d_steer = s.out(steer, e); double* d_speed = s.out(speed, e); int32_t* d_bi = s.out(best_idx, e);
    double* d_bc = s.out(best_cost, e); double* d_bs = s.out(best_seq, e * T * 2);
    if ((rc = launch_stmpc_shoot(ctx, d_x0, d_ref, d_c, E, cfg, d_steer, d_speed, d_bi, d_bc, d_bs))) return rc;
    return s.finish();
}

}  // extern "C"
#pragma GCC visibility pop
