// gnna_runtime.hip -- per-device runtime state of libgnna.so: CU count, the ring of
// "partition is not canonical" flags, per-stream scratch buffers, and the optional HIP-event
// timing of the kernels of each aggregation call (gnna_profile_begin / gnna_profile_end).
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "gnna.h"
#include "gnna_internal.h"

namespace gnna {

namespace {
constexpr int kMaxDevices = 64;
DeviceState g_dev[kMaxDevices];
std::mutex g_dev_mutex;
std::atomic<uint32_t> g_seq{0};

}  // namespace

int get_device_state(DeviceState **out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(GNNA_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return fail(GNNA_ERR_UNSUPPORTED, "device ordinal %d", dev);
    DeviceState &s = g_dev[dev];
    if (!s.init.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lock(g_dev_mutex);
        if (!s.init.load(std::memory_order_relaxed)) {
            hipDeviceProp_t prop;
            e = hipGetDeviceProperties(&prop, dev);
            if (e != hipSuccess)
                return fail(GNNA_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
            s.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            e = hipMalloc(reinterpret_cast<void **>(&s.flags), 2 * kFlagSlots * sizeof(int32_t));
            if (e != hipSuccess) return fail(GNNA_ERR_HIP, "hipMalloc(flags): %s", hipGetErrorString(e));
            e = hipMemset(s.flags, 0, 2 * kFlagSlots * sizeof(int32_t));
            if (e != hipSuccess) return fail(GNNA_ERR_HIP, "hipMemset(flags): %s", hipGetErrorString(e));
            const size_t gap_bytes = (size_t)kCallBlocks * kGapWords * sizeof(unsigned long long);
            e = hipMalloc(reinterpret_cast<void **>(&s.gap_lists), gap_bytes);
            if (e != hipSuccess) return fail(GNNA_ERR_HIP, "hipMalloc(gap lists): %s", hipGetErrorString(e));
            e = hipMemset(s.gap_lists, 0, gap_bytes);
            if (e != hipSuccess) return fail(GNNA_ERR_HIP, "hipMemset(gap lists): %s", hipGetErrorString(e));
            const size_t sync_bytes = (size_t)kCallBlocks * kSweepSlotWords * sizeof(uint32_t);
            e = hipMalloc(reinterpret_cast<void **>(&s.sweep_sync), sync_bytes);
            if (e != hipSuccess) return fail(GNNA_ERR_HIP, "hipMalloc(sweep counters): %s", hipGetErrorString(e));
            e = hipMemset(s.sweep_sync, 0, sync_bytes);
            if (e != hipSuccess) return fail(GNNA_ERR_HIP, "hipMemset(sweep counters): %s", hipGetErrorString(e));
            s.init.store(true, std::memory_order_release);
        }
    }
    *out = &s;
    return GNNA_OK;
}

// Grow-only scratch buffer for `stream` (work on one stream is ordered, so one buffer per
// stream is enough).  hipFree of the old buffer synchronises the device, which makes the
// replacement safe; steady state performs no allocation.
int get_workspace(DeviceState *ds, hipStream_t stream, int slot, size_t bytes, void **out)
{
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    Workspace &w = ds->ws[std::make_pair(stream, slot)];
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
    if (w.bytes < bytes) {
        // growing means hipFree + hipMalloc: illegal inside a stream capture -- refuse instead (warm the path up before
        // capturing on this stream)
        if (cap != hipStreamCaptureStatusNone)
            return fail(GNNA_ERR_UNSUPPORTED, "library scratch (%zu B) would have to grow during stream capture: "
                        "run the same call once on this stream before capturing", bytes);
        // a buffer some captured graph points at stays allocated for good (the graph may be replayed at any time)
        if (w.ptr && !w.captured) (void)hipFree(w.ptr);
        w.ptr = nullptr;
        w.bytes = 0;
        w.captured = false;
        const size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(&w.ptr, want);
        if (e != hipSuccess) return fail(GNNA_ERR_HIP, "hipMalloc(workspace %zu B): %s", want, hipGetErrorString(e));
        w.bytes = want;
    }
    if (cap != hipStreamCaptureStatusNone) w.captured = true;
    *out = w.ptr;
    return GNNA_OK;
}

namespace {
thread_local int32_t t_block_seq = 0;     // the calling thread's current call and the block it was given (a call asks several
thread_local int t_block = 0;             // times: flags, gap list, sweep counters -- and must get the same answer)

int pick_call_block(DeviceState *ds, hipStream_t stream, int32_t seq)
{
    // (seq is odd and unique per call, next_call_seq: its upper bits count the calls, so the ring uses every block)
    const int ring0 = kStreamBlocks + kCapturedBlocks;
    const int shared = ring0 + (int)(((uint32_t)seq >> 1) % (uint32_t)(kCallBlocks - ring0));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    if (cap != hipStreamCaptureStatusNone)
        return ds->captured_blocks < kCapturedBlocks ? kStreamBlocks + ds->captured_blocks++ : shared;
    auto it = ds->stream_block.find(stream);
    if (it != ds->stream_block.end()) return it->second;
    if ((int)ds->stream_block.size() < kStreamBlocks) {
        const int b = (int)ds->stream_block.size();
        ds->stream_block.emplace(stream, b);
        return b;
    }
    return shared;
}
}  // namespace

int call_block_of(DeviceState *ds, hipStream_t stream, int32_t seq)
{
    if (seq != t_block_seq) { t_block = pick_call_block(ds, stream, seq); t_block_seq = seq; }
    return t_block;
}

int32_t next_call_seq(DeviceState *ds, hipStream_t stream, int32_t **flag_slot)
{
    const uint32_t seq_u = g_seq.fetch_add(1) + 1;
    const int32_t seq = (int32_t)(((seq_u << 1) | 1u) & 0x7fffffffu);  // never 0, and different for consecutive calls: a flag raised
                                                                       // by one call is never mistaken for the next one's on the same stream
    // the flags are compared with the call's own sequence number, so a slot can be shared by calls that follow each other on one
    // stream; calls on different streams use different slots (call_block_of), so none can overwrite the flag of another in flight
    *flag_slot = ds->flags + call_block_of(ds, stream, seq);
    return seq;
}

namespace {
struct ProfileState {
    std::atomic<bool> on{false};
    int max_calls = 0;
    int calls = 0;
    std::vector<hipEvent_t> ev;  // 3 per call: before prologue, between, after main
};
ProfileState g_prof;
std::mutex g_prof_mutex;

hipEvent_t prof_event(int call, int which)
{
    return g_prof.ev[(size_t)call * 3 + which];
}

}  // namespace

int profile_acquire_call(bool has_work)
{
    if (!g_prof.on) return -1;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (g_prof.on && g_prof.calls < g_prof.max_calls && has_work) return g_prof.calls++;
    return -1;
}

void profile_record(int call, int which, hipStream_t stream)
{
    if (call >= 0) (void)hipEventRecord(prof_event(call, which), stream);
}

}  // namespace gnna

using namespace gnna;

extern "C" {
#pragma GCC visibility push(default)

int gnna_profile_begin(int max_calls)
{
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (g_prof.on) return fail(GNNA_ERR_INVALID_ARGUMENT, "profiling already active");
    if (max_calls <= 0 || max_calls > (1 << 20))
        return fail(GNNA_ERR_INVALID_ARGUMENT, "max_calls out of range: %d", max_calls);
    g_prof.ev.resize((size_t)max_calls * 3);
    for (auto &e : g_prof.ev) {
        hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) return fail(GNNA_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(rc));
    }
    g_prof.max_calls = max_calls;
    g_prof.calls = 0;
    g_prof.on = true;
    return GNNA_OK;
}

int gnna_profile_end(double *avg_main_ms, double *avg_prologue_ms, int *num_calls)
{
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (!g_prof.on) return fail(GNNA_ERR_INVALID_ARGUMENT, "profiling not active");
    g_prof.on = false;
    double main_ms = 0, pro_ms = 0;
    int rc_out = GNNA_OK;
    for (int c = 0; c < g_prof.calls; c++) {
        hipError_t rc = hipEventSynchronize(prof_event(c, 2));
        float a = 0, b = 0;
        if (rc == hipSuccess) rc = hipEventElapsedTime(&a, prof_event(c, 0), prof_event(c, 1));
        if (rc == hipSuccess) rc = hipEventElapsedTime(&b, prof_event(c, 1), prof_event(c, 2));
        if (rc != hipSuccess) { rc_out = fail(GNNA_ERR_HIP, "profile events: %s", hipGetErrorString(rc)); break; }
        pro_ms += a;
        main_ms += b;
    }
    const int n = g_prof.calls;
    for (auto &e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear();
    g_prof.calls = 0;
    if (num_calls) *num_calls = n;
    if (avg_main_ms) *avg_main_ms = n ? main_ms / n : 0.0;
    if (avg_prologue_ms) *avg_prologue_ms = n ? pro_ms / n : 0.0;
    return rc_out;
}

int gnna_device_cus(void)
{
    int n = 0, dev = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 0; }
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return cus;
}
#pragma GCC visibility pop
}  // extern "C"
