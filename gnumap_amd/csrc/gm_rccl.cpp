// gm_rccl.cpp — multi-GPU combination of the coverage track: the reference's MPI Allreduce(SUM) of amount_genome
// (src/Driver.cpp:1660-1672, 64 Mi-float chunks) becomes ONE ncclAllReduce over xGMI on the device-resident arrays.
// Single-process form (one gm_index per GPU); a torch.distributed / one-process-per-GPU caller instead all-reduces
// gm_coverage_device_ptr() itself (bench.py does that through RCCL).
#include "gm_host.h"
#include <rccl/rccl.h>
#include <vector>

extern "C" int gm_coverage_allreduce(gm_index** per_gpu, int n_gpu) {
    if (!per_gpu || n_gpu <= 0) return GM_E_ARG;
    if (n_gpu == 1) return GM_OK;
    std::vector<int> devs((size_t)n_gpu);
    std::vector<void*> ptrs((size_t)n_gpu);
    uint64_t bins = gm_coverage_bins(per_gpu[0]);
    gm_index_info info;
    for (int i = 0; i < n_gpu; ++i) {
        if (!per_gpu[i] || gm_coverage_bins(per_gpu[i]) != bins || bins == 0) { gm_set_error("coverage tracks differ between GPUs"); return GM_E_ARG; }
        gm_index_get_info(per_gpu[i], &info);
        devs[(size_t)i] = info.device_id;
        ptrs[(size_t)i] = gm_coverage_device_ptr(per_gpu[i]);
    }
    std::vector<ncclComm_t> comms((size_t)n_gpu);
    if (ncclCommInitAll(comms.data(), n_gpu, devs.data()) != ncclSuccess) { gm_set_error("ncclCommInitAll failed"); return GM_E_HIP; }
    int rc = GM_OK;
    ncclGroupStart();
    for (int i = 0; i < n_gpu; ++i) {
        (void)hipSetDevice(devs[(size_t)i]);
        if (ncclAllReduce(ptrs[(size_t)i], ptrs[(size_t)i], bins, ncclFloat, ncclSum, comms[(size_t)i], nullptr) != ncclSuccess) rc = GM_E_HIP;
    }
    ncclGroupEnd();
    if (gm_coverage_nuc_device_ptr(per_gpu[0])) {           // -b / -d: the five per-nucleotide arrays (src/Driver.cpp:1719-1768)
        ncclGroupStart();
        for (int i = 0; i < n_gpu; ++i) {
            (void)hipSetDevice(devs[(size_t)i]);
            void* q = gm_coverage_nuc_device_ptr(per_gpu[i]);
            if (!q || ncclAllReduce(q, q, 5 * bins, ncclFloat, ncclSum, comms[(size_t)i], nullptr) != ncclSuccess) rc = GM_E_HIP;
        }
        ncclGroupEnd();
    }
    for (int i = 0; i < n_gpu; ++i) {
        (void)hipSetDevice(devs[(size_t)i]);
        if (hipDeviceSynchronize() != hipSuccess) rc = GM_E_HIP;
        ncclCommDestroy(comms[(size_t)i]);
    }
    if (rc) gm_set_error("ncclAllReduce of the coverage track failed");
    return rc;
}
