// gm_rccl.cpp — multi-GPU combination of the coverage track: the reference's MPI Allreduce(SUM) of amount_genome
// (src/Driver.cpp:1660-1672, 64 Mi-float chunks) and the Reduce of the per-nucleotide arrays (src/Driver.cpp:1719-1768) become
// in-place ncclAllReduce calls over xGMI on the device-resident arrays.  Single-process form (one gm_index per GPU, what the
// gnumap driver uses with --gpus N); a one-process-per-GPU caller (bench.py under torch.distributed) all-reduces
// gm_coverage_device_ptr() itself.
// Every RCCL call is checked; the first failure aborts the communicators (ncclCommAbort) instead of issuing further work.  Each
// device gets its own non-blocking stream.  GM_RCCL_FORCE=1 takes the RCCL path for a single GPU too (a one-rank communicator),
// so the code can be rehearsed on a one-GPU box.
#include "gm_host.h"
#include <rccl/rccl.h>
#include <cstdlib>
#include <string>
#include <vector>

extern "C" int gm_coverage_allreduce(gm_index** per_gpu, int n_gpu) {
    if (!per_gpu || n_gpu <= 0) return GM_E_ARG;
    const char* force = getenv("GM_RCCL_FORCE");
    if (n_gpu == 1 && !(force && atoi(force))) return GM_OK;
    std::vector<int> devs((size_t)n_gpu);
    std::vector<void*> ptrs((size_t)n_gpu), nucs((size_t)n_gpu);
    const uint64_t bins = gm_coverage_bins(per_gpu[0]);
    gm_index_info info;
    bool any_nuc = false;
    for (int i = 0; i < n_gpu; ++i) {
        if (!per_gpu[i] || gm_coverage_bins(per_gpu[i]) != bins || bins == 0) { gm_set_error("coverage tracks differ between GPUs (or gm_coverage_reset was not called)"); return GM_E_ARG; }
        gm_index_get_info(per_gpu[i], &info);
        devs[(size_t)i] = info.device_id;
        ptrs[(size_t)i] = gm_coverage_device_ptr(per_gpu[i]);
        nucs[(size_t)i] = gm_coverage_nuc_device_ptr(per_gpu[i]);
        any_nuc |= nucs[(size_t)i] != nullptr;
        for (int j = 0; j < i; ++j)
            if (devs[(size_t)j] == devs[(size_t)i]) { gm_set_error("gm_coverage_allreduce: two indexes on the same device"); return GM_E_ARG; }
    }
    if (any_nuc) for (int i = 0; i < n_gpu; ++i) if (!nucs[(size_t)i]) { gm_set_error("per-nucleotide track enabled on some GPUs only"); return GM_E_ARG; }

    std::vector<ncclComm_t> comms((size_t)n_gpu, nullptr);
    std::vector<hipStream_t> streams((size_t)n_gpu, nullptr);
    std::string err;
    auto nccl_ok = [&](ncclResult_t r, const char* what) {
        if (r == ncclSuccess) return true;
        if (err.empty()) err = std::string(what) + ": " + ncclGetErrorString(r);
        return false;
    };
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        if (err.empty()) err = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    };
    bool ok = nccl_ok(ncclCommInitAll(comms.data(), n_gpu, devs.data()), "ncclCommInitAll");
    for (int i = 0; ok && i < n_gpu; ++i)
        ok = hip_ok(hipSetDevice(devs[(size_t)i]), "hipSetDevice") && hip_ok(hipStreamCreateWithFlags(&streams[(size_t)i], hipStreamNonBlocking), "hipStreamCreate");
    // one group per array: the track itself, then (with -b / -d) the five per-nucleotide arrays as one 5 x bins buffer
    auto reduce_all = [&](const std::vector<void*>& bufs, uint64_t count, const char* what) {
        if (!nccl_ok(ncclGroupStart(), "ncclGroupStart")) return false;
        bool issued = true;
        for (int i = 0; issued && i < n_gpu; ++i)
            issued = hip_ok(hipSetDevice(devs[(size_t)i]), "hipSetDevice") &&
                     nccl_ok(ncclAllReduce(bufs[(size_t)i], bufs[(size_t)i], count, ncclFloat, ncclSum, comms[(size_t)i], streams[(size_t)i]), what);
        // the group has to be closed even when a call inside it failed; its own result counts too
        const bool ended = nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
        return issued && ended;
    };
    if (ok) ok = reduce_all(ptrs, bins, "ncclAllReduce(coverage track)");
    if (ok && any_nuc) ok = reduce_all(nucs, 5 * bins, "ncclAllReduce(per-nucleotide track)");
    for (int i = 0; ok && i < n_gpu; ++i)
        ok = hip_ok(hipSetDevice(devs[(size_t)i]), "hipSetDevice") && hip_ok(hipStreamSynchronize(streams[(size_t)i]), "hipStreamSynchronize");
    for (int i = 0; ok && i < n_gpu; ++i) {                  // asynchronous errors surface here
        ncclResult_t ar = ncclSuccess;
        ok = nccl_ok(ncclCommGetAsyncError(comms[(size_t)i], &ar), "ncclCommGetAsyncError") && nccl_ok(ar, "asynchronous RCCL error");
    }
    for (int i = 0; i < n_gpu; ++i) {
        (void)hipSetDevice(devs[(size_t)i]);
        if (comms[(size_t)i]) { if (ok) (void)ncclCommDestroy(comms[(size_t)i]); else (void)ncclCommAbort(comms[(size_t)i]); }
        if (streams[(size_t)i]) (void)hipStreamDestroy(streams[(size_t)i]);
    }
    if (!ok) { gm_set_error("gm_coverage_allreduce: " + err); return GM_E_HIP; }
    return GM_OK;
}
