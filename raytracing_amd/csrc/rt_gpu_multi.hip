/*
 * rt_gpu_multi.hip -- one frame over several GPUs of a node, from a single-process C host.
 *
 * The host program renders a frame with 15 threads pulling tiles from a queue (reference:
 * src/main.c:1056-1074, thread_entry_point src/main.c:804-833).  Here the frame is cut into
 * 8-row bands dealt round-robin to the devices; every device renders its bands into a packed
 * buffer (rt_gpu_render_region_device with RT_GPU_FLAG_COMPACT_BANDS), one RCCL all-gather over
 * xGMI leaves every device with all bands, and a small kernel puts them back in image order.
 * Nothing is exchanged while rendering: the random streams are keyed by (seed, pixel, sample),
 * so the frame is the single-device frame bit for bit.
 *
 * RCCL is taken from the process at run time (dlopen), so librt_gpu.so has no link-time
 * dependency on it; one device needs no RCCL at all.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rt_gpu.h"

extern "C" int rt_gpu_set_error(int code, const char *what);     /* rt_gpu.hip: records the calling thread's error text */

namespace {

/* the few RCCL entry points used (rccl.h: ncclCommInitAll, ncclAllGather, ncclGroupStart/End, ncclCommDestroy) */
typedef void *nccl_comm;
struct Rccl {
    void *lib = nullptr;
    int (*comm_init_all)(nccl_comm *, int, const int *) = nullptr;
    int (*comm_destroy)(nccl_comm) = nullptr;
    int (*all_gather)(const void *, void *, size_t, int, nccl_comm, hipStream_t) = nullptr;
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    const char *(*error_string)(int) = nullptr;
    std::mutex guard;
    bool load()                      /* any thread, any number of times: the first call loads, the others find it loaded */
    {
        std::lock_guard<std::mutex> hold(guard);
        if (lib) return comm_init_all && comm_destroy && all_gather && group_start && group_end;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
        comm_init_all = reinterpret_cast<decltype(comm_init_all)>(dlsym(lib, "ncclCommInitAll"));
        comm_destroy = reinterpret_cast<decltype(comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
        all_gather = reinterpret_cast<decltype(all_gather)>(dlsym(lib, "ncclAllGather"));
        group_start = reinterpret_cast<decltype(group_start)>(dlsym(lib, "ncclGroupStart"));
        group_end = reinterpret_cast<decltype(group_end)>(dlsym(lib, "ncclGroupEnd"));
        error_string = reinterpret_cast<decltype(error_string)>(dlsym(lib, "ncclGetErrorString"));
        return comm_init_all && comm_destroy && all_gather && group_start && group_end;
    }
};
Rccl g_rccl;
const int NCCL_FLOAT = 7;            /* ncclFloat32 in rccl.h's ncclDataType_t */

struct Device {
    int id = 0;
    rt_gpu_scene *scene = nullptr;
    hipStream_t stream = nullptr;
    float *mine = nullptr, *all = nullptr, *frame = nullptr;
    size_t mine_floats = 0, frame_floats = 0;
    nccl_comm comm = nullptr;
};

}  // namespace

struct rt_gpu_multi {
    std::vector<Device> dev;
    uint32_t band_rows = 8;
    bool collective = false;         /* the bands go through RCCL (more than one device, or RT_GPU_MULTI_FORCE_RCCL) */
};

/* image row y lives in band y / band of rank (band index) % world, at packed row (band index / world) * band + y % band */
__global__ void rt_assemble_bands_kernel(const float *all, float *frame, uint32_t width, uint32_t height,
                                         uint32_t band, uint32_t world, uint32_t rows_per_rank)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row_floats = (size_t)width * 3;
    if (i >= row_floats * height) return;
    const uint32_t y = (uint32_t)(i / row_floats);
    const uint32_t b = y / band, r = b % world, k = b / world;
    const size_t src_row = (size_t)r * rows_per_rank + (size_t)k * band + y % band;
    frame[i] = all[src_row * row_floats + i % row_floats];
}

extern "C" {

uint32_t rt_gpu_multi_rows_per_rank(uint32_t height, uint32_t band_rows, uint32_t world)
{
    if (!band_rows || !world) return 0;
    const uint32_t bands = (height + band_rows - 1) / band_rows;
    return ((bands + world - 1) / world) * band_rows;
}

int rt_gpu_assemble_bands_device(const float *device_all, float *device_frame, uint32_t width, uint32_t height,
                                 uint32_t band_rows, uint32_t world, void *hip_stream)
{
    if (!device_all || !device_frame || !width || !height || !band_rows || !world)
        return rt_gpu_set_error(RT_GPU_ERR_ARGUMENT, "rt_gpu_assemble_bands_device: bad argument");
    const size_t n = (size_t)width * height * 3;
    hipLaunchKernelGGL(rt_assemble_bands_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                       device_all, device_frame, width, height, band_rows, world, rt_gpu_multi_rows_per_rank(height, band_rows, world));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rt_gpu_set_error(RT_GPU_ERR_DEVICE, hipGetErrorString(e));
    return RT_GPU_OK;
}

void rt_gpu_multi_destroy(rt_gpu_multi *m)
{
    if (!m) return;
    for (auto &d : m->dev) {
        (void)hipSetDevice(d.id);
        (void)hipDeviceSynchronize();
        if (d.comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(d.comm);
        if (d.scene) rt_gpu_scene_destroy(d.scene);
        if (d.mine) (void)hipFree(d.mine);
        if (d.all) (void)hipFree(d.all);
        if (d.frame) (void)hipFree(d.frame);
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    delete m;
}

int rt_gpu_multi_create(const rt_scene *scene, int device_count, rt_gpu_multi **out)
{
    if (!scene || !out) return rt_gpu_set_error(RT_GPU_ERR_ARGUMENT, "rt_gpu_multi_create: NULL argument");
    *out = nullptr;
    const int visible = rt_gpu_device_count();
    if (visible <= 0) return rt_gpu_set_error(RT_GPU_ERR_NO_DEVICE, "no HIP device is visible");
    if (device_count <= 0) device_count = visible;
    if (device_count > visible)
        return rt_gpu_set_error(RT_GPU_ERR_ARGUMENT, ("rt_gpu_multi_create: " + std::to_string(device_count) + " devices asked for, " +
                                                       std::to_string(visible) + " visible").c_str());
    int before = 0;
    (void)hipGetDevice(&before);
    rt_gpu_multi *m = new rt_gpu_multi();
    m->dev.resize((size_t)device_count);
    int rc = 0;
    for (int i = 0; i < device_count && !rc; i++) {
        Device &d = m->dev[(size_t)i];
        d.id = i;
        if (hipSetDevice(i) != hipSuccess || hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess)
            rc = rt_gpu_set_error(RT_GPU_ERR_DEVICE, "rt_gpu_multi_create: cannot use a device");
        else rc = rt_gpu_scene_create(scene, &d.scene);            /* the scene resident on every device */
    }
    /* RT_GPU_MULTI_FORCE_RCCL: the collective also with one device (a communicator of one rank), so that a
     * one-GPU machine can check the library loading and the calls */
    m->collective = device_count > 1 || getenv("RT_GPU_MULTI_FORCE_RCCL") != nullptr;
    if (!rc && m->collective) {
        if (!g_rccl.load()) rc = rt_gpu_set_error(RT_GPU_ERR_DEVICE, "rt_gpu_multi_create: librccl.so could not be loaded (needed for more than one device)");
        else {
            std::vector<nccl_comm> comms((size_t)device_count);
            std::vector<int> ids((size_t)device_count);
            for (int i = 0; i < device_count; i++) ids[(size_t)i] = i;
            const int e = g_rccl.comm_init_all(comms.data(), device_count, ids.data());
            if (e) rc = rt_gpu_set_error(RT_GPU_ERR_DEVICE, (std::string("ncclCommInitAll: ") + (g_rccl.error_string ? g_rccl.error_string(e) : "failed")).c_str());
            else for (int i = 0; i < device_count; i++) m->dev[(size_t)i].comm = comms[(size_t)i];
        }
    }
    (void)hipSetDevice(before);
    if (rc) { rt_gpu_multi_destroy(m); return rc; }
    *out = m;
    return RT_GPU_OK;
}

int rt_gpu_multi_device_count(const rt_gpu_multi *m) { return m ? (int)m->dev.size() : 0; }

int rt_gpu_multi_render_frame(rt_gpu_multi *m, rt_image3f *image, uint32_t spp, uint64_t seed, const rt_gpu_options *options)
{
    if (!m || !image || !image->pixels) return rt_gpu_set_error(RT_GPU_ERR_ARGUMENT, "rt_gpu_multi_render_frame: NULL argument");
    if (options && options->struct_size != sizeof(rt_gpu_options)) return rt_gpu_set_error(RT_GPU_ERR_ARGUMENT, "rt_gpu_options.struct_size mismatch");
    if (options && (options->band_rows || options->frames > 1 || (options->flags & (RT_GPU_FLAG_COMPACT_BANDS | RT_GPU_FLAG_ACCUMULATE))))
        return rt_gpu_set_error(RT_GPU_ERR_ARGUMENT, "rt_gpu_multi_render_frame splits the frame itself: no bands, frames or accumulation in the options");
    const uint32_t W = image->size.x, H = image->size.y, world = (uint32_t)m->dev.size(), band = m->band_rows;
    if (!W || !H || !spp) return rt_gpu_set_error(RT_GPU_ERR_ARGUMENT, "empty image or sample count");
    const uint32_t rows = rt_gpu_multi_rows_per_rank(H, band, world);
    const size_t mine_floats = (size_t)rows * W * 3, frame_floats = (size_t)W * H * 3;
    int before = 0;
    (void)hipGetDevice(&before);
    int rc = 0;
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !rc) rc = rt_gpu_set_error(RT_GPU_ERR_DEVICE, (std::string(what) + ": " + hipGetErrorString(e)).c_str());
        return e == hipSuccess;
    };
    rt_rect2u whole;
    whole.point.x = 0; whole.point.y = 0; whole.size.x = W; whole.size.y = H;
    for (uint32_t i = 0; i < world && !rc; i++) {                   /* every device starts on its bands */
        Device &d = m->dev[i];
        if (!hip_ok(hipSetDevice(d.id), "hipSetDevice")) break;
        if (d.mine_floats < mine_floats) {
            if (d.mine) (void)hipFree(d.mine);
            if (d.all) (void)hipFree(d.all);
            d.mine = d.all = nullptr; d.mine_floats = 0;
            if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d.mine), mine_floats * 4), "hipMalloc") ||
                !hip_ok(hipMalloc(reinterpret_cast<void **>(&d.all), mine_floats * 4 * world), "hipMalloc")) break;
            d.mine_floats = mine_floats;
        }
        if (i == 0 && d.frame_floats < frame_floats) {
            if (d.frame) (void)hipFree(d.frame);
            d.frame = nullptr; d.frame_floats = 0;
            if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d.frame), frame_floats * 4), "hipMalloc")) break;
            d.frame_floats = frame_floats;
        }
        if (!hip_ok(hipMemsetAsync(d.mine, 0, mine_floats * 4, d.stream), "hipMemsetAsync")) break;   /* a short last band leaves padding rows */
        rt_gpu_options o;
        memset(&o, 0, sizeof(o));
        if (options) o = *options;
        o.struct_size = sizeof(o);
        o.band_rows = band; o.band_rank = i; o.band_count = world;
        o.flags |= RT_GPU_FLAG_COMPACT_BANDS;
        rc = rt_gpu_render_region_device(d.scene, d.mine, W, H, whole, spp, seed, &o, d.stream);
    }
    if (!rc) {                                                        /* every device receives every device's bands */
        if (m->collective) {
            int e = g_rccl.group_start();
            for (uint32_t i = 0; i < world && !e; i++) {
                Device &d = m->dev[i];
                (void)hipSetDevice(d.id);
                e = g_rccl.all_gather(d.mine, d.all, mine_floats, NCCL_FLOAT, d.comm, d.stream);
            }
            const int e2 = g_rccl.group_end();
            if (e || e2) rc = rt_gpu_set_error(RT_GPU_ERR_DEVICE, (std::string("ncclAllGather: ") + (g_rccl.error_string ? g_rccl.error_string(e ? e : e2) : "failed")).c_str());
        } else {
            Device &d = m->dev[0];
            (void)hipSetDevice(d.id);
            hip_ok(hipMemcpyAsync(d.all, d.mine, mine_floats * 4, hipMemcpyDeviceToDevice, d.stream), "hipMemcpyAsync");
        }
    }
    if (!rc) {                                                        /* device 0 puts the bands in image order and hands the frame to the host */
        Device &d = m->dev[0];
        (void)hipSetDevice(d.id);
        rc = rt_gpu_assemble_bands_device(d.all, d.frame, W, H, band, world, d.stream);
        if (!rc) hip_ok(hipMemcpyAsync(&image->pixels[0].x, d.frame, frame_floats * 4, hipMemcpyDeviceToHost, d.stream), "hipMemcpyAsync");
    }
    for (uint32_t i = 0; i < world; i++) {                            /* also on failure: leave no work behind */
        (void)hipSetDevice(m->dev[i].id);
        hipError_t e = hipStreamSynchronize(m->dev[i].stream);
        if (!rc) hip_ok(e, "hipStreamSynchronize");
    }
    (void)hipSetDevice(before);
    return rc;
}

} /* extern "C" */
