// psfm_ingest.hip -- point_trajectory/utils.py:26-56 (load_flows / read_flo) for a whole stack, straight into HBM.
//
// A .flo file (Middlebury): 12-byte header {f32 magic 202021.25, i32 width, i32 height} + h * w interleaved (u, v) f32 pairs -- the
// layout the kernels read.  The Python mirror's ingest (reader threads -> pinned staging -> async H2D) moves a 1080p stack at 30-40 GB/s
// but pays ~100 us of interpreter work per FILE, which is what a stack of small frames is made of: 2 x 49 files of 3.3 MB (a DAVIS-sized
// sequence) took 15-18 ms = 18-21 GB/s, two thirds of that sequence's whole disk-to-disk time (profiles/r05/r05_k_e2e_small.txt).  Here the
// same pipeline without the interpreter: `n_threads` reader threads take files in order, read() each into its slot of a ring of pinned
// buffers owned by the context (a slot is reused once the copy that last read it has completed), and enqueue the H2D copy on the
// context's copy stream themselves; the caller's stream waits for the last copy.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "psfm_internal.h"

namespace {
struct FloHeader { float magic; int32_t w, h; };
static_assert(sizeof(FloHeader) == 12, ".flo header");
}  // namespace

extern "C" psfm_status psfm_load_flo_stack(psfm_ctx* c, const char* const* paths, int n, int h, int w, float* dst_dev, int n_threads, void* stream)
{
    if (!c || n < 0 || h < 1 || w < 1 || (n > 0 && (!paths || !dst_dev))) { psfm_set_error("psfm_load_flo_stack: bad argument (n=%d h=%d w=%d)", n, h, w); return PSFM_ERR_ARG; }
    if (n == 0) return PSFM_OK;
    PSFM_HIP(hipSetDevice(c->device));
    PsfmGate gate(c->device, 0);
    const size_t bytes = (size_t)h * w * 2 * sizeof(float);
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 32) n_threads = 32;
    if (n_threads > n) n_threads = n;
    // ---- the ring of pinned staging buffers (grow-only, owned by the context) ----
    int n_slots = 2 * n_threads;
    while (n_slots > 2 && (size_t)n_slots * bytes > ((size_t)1 << 30)) --n_slots;       // (at most 1 GiB of pinned memory)
    if (n_slots > n) n_slots = n;
    if (c->ingest_slot_bytes < bytes || (int)c->ingest_slots.size() < n_slots) {
        for (void* p : c->ingest_slots) (void)hipHostFree(p);
        c->ingest_slots.clear();
        c->ingest_slot_bytes = 0;
        for (int k = 0; k < n_slots; ++k) {
            void* p = nullptr;
            if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { psfm_set_error("psfm_load_flo_stack: hipHostMalloc(%zu) failed", bytes); return PSFM_ERR_HIP; }
            c->ingest_slots.push_back(p);
        }
        c->ingest_slot_bytes = bytes;
    }
    n_slots = (int)c->ingest_slots.size() < n_slots ? (int)c->ingest_slots.size() : n_slots;
    if (!c->copy_stream) PSFM_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    while ((int)c->ingest_events.size() < n_slots) {
        hipEvent_t e;
        PSFM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->ingest_events.push_back(e);
    }
    // two copy streams, the slots alternating between them: the runtime gives streams their own SDMA engine, and one engine moves
    // ~35 GB/s of a PCIe Gen5 x16 link's ~55 (PSFM_FLO_COPY_STREAMS=1: one stream, as in round 5)
    static const int n_copy_env = getenv("PSFM_FLO_COPY_STREAMS") ? atoi(getenv("PSFM_FLO_COPY_STREAMS")) : 2;
    const bool two = n_copy_env >= 2;
    if (two && !c->copy_stream2) PSFM_HIP(hipStreamCreateWithFlags(&c->copy_stream2, hipStreamNonBlocking));
    hipStream_t cs = c->copy_stream;
    hipStream_t cs2 = two ? c->copy_stream2 : c->copy_stream;
    hipStream_t s = (hipStream_t)stream;
    {   // the copies start behind whatever the caller has enqueued on `stream` (e.g. the allocation of dst on that stream)
        hipEvent_t e = c->prof.get();
        PSFM_HIP(hipEventRecord(e, s));
        PSFM_HIP(hipStreamWaitEvent(cs, e, 0));
        if (two) PSFM_HIP(hipStreamWaitEvent(cs2, e, 0));
        c->prof.pool.push_back(e);
    }
    // ---- the readers ----
    std::atomic<int> next(0);
    std::atomic<bool> failed(false);
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> enqueued((size_t)n_slots, -1);      // last file whose copy out of slot k has been enqueued (-1: the slot is fresh)
    std::string err;
    const int device = c->device;
    auto fail = [&](const std::string& m) {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed.exchange(true)) err = m;
        cv.notify_all();
    };
    auto worker = [&]() {
        if (hipSetDevice(device) != hipSuccess) { fail("hipSetDevice failed"); return; }
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n || failed.load()) return;
            const int k = i % n_slots;
            if (i >= n_slots) {      // the slot's previous tenant: its copy must have been enqueued, then completed
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return enqueued[(size_t)k] == i - n_slots || failed.load(); });
                if (failed.load()) return;
                lk.unlock();
                if (hipEventSynchronize(c->ingest_events[(size_t)k]) != hipSuccess) { fail("hipEventSynchronize failed"); return; }
            }
            const char* path = paths[i];
            const int fd = open(path, O_RDONLY);
            if (fd < 0) { fail(std::string(path) + ": " + strerror(errno)); return; }
            FloHeader hd;
            ssize_t got = read(fd, &hd, sizeof(hd));
            if (got != (ssize_t)sizeof(hd) || hd.magic != 202021.25f) { close(fd); fail(std::string(path) + " does not start with the .flo magic number 202021.25"); return; }
            if (hd.w != w || hd.h != h) {
                close(fd);
                fail(std::string(path) + ": frame size " + std::to_string(hd.w) + " x " + std::to_string(hd.h) + " differs from the stack's " + std::to_string(w) + " x " + std::to_string(h));
                return;
            }
            char* dst = (char*)c->ingest_slots[(size_t)k];
            size_t have = 0;
            while (have < bytes) {
                got = read(fd, dst + have, bytes - have);
                if (got < 0 && errno == EINTR) continue;
                if (got <= 0) break;
                have += (size_t)got;
            }
            close(fd);
            if (have != bytes) { fail(std::string(path) + ": truncated (" + std::to_string(have) + " of " + std::to_string(bytes) + " data bytes)"); return; }
            {   // copies are enqueued in any order (each waits only for its own slot); the event tells when the slot is free again
                std::lock_guard<std::mutex> lk(mu);
                hipStream_t q = (k & 1) ? cs2 : cs;
                if (hipMemcpyAsync((char*)dst_dev + (size_t)i * bytes, dst, bytes, hipMemcpyHostToDevice, q) != hipSuccess ||
                    hipEventRecord(c->ingest_events[(size_t)k], q) != hipSuccess) {
                    if (!failed.exchange(true)) err = "hipMemcpyAsync / hipEventRecord failed";
                    cv.notify_all();
                    return;
                }
                enqueued[(size_t)k] = i;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> ths;
    // (a thread that cannot be created -- std::system_error -- must not unwind through the extern "C" entry point with joinable threads
    // behind it: the workers pull frames from one queue, so the ones that did start, or this thread, read everything)
    for (int t = 0; t < n_threads; ++t) {
        try { ths.emplace_back(worker); } catch (const std::exception&) { break; }
    }
    if (ths.empty()) worker();
    for (auto& t : ths) t.join();
    // the staging buffers belong to the context: nothing may still read them when the call returns; `stream` continues behind the copies
    hipError_t e1 = hipStreamSynchronize(cs);
    if (two) { const hipError_t e2 = hipStreamSynchronize(cs2); if (e1 == hipSuccess) e1 = e2; }
    if (failed.load()) { psfm_set_error("psfm_load_flo_stack: %s", err.c_str()); return PSFM_ERR_ARG; }
    PSFM_HIP(e1);
    return PSFM_OK;
}
