// gm_host.h — host-side structures of libgnumap_hip (not part of the public ABI)
#pragma once
#include "../../include/gnumap_hip.h"
#include <hip/hip_runtime.h>
#include "gm_internal.h"
#include <string>
#include <vector>

struct GmContig { std::string name; uint64_t offset; uint32_t len; };

struct GmHostIndex {
    uint64_t primary = 0, L2[5] = { 0, 0, 0, 0, 0 }, seq_len = 0, bwt_words = 0, n_sa = 0, l_pac = 0;
    uint32_t sa_intv = 32;
    std::vector<uint32_t> bwt;
    std::vector<uint64_t> sa;
    std::vector<uint8_t> pac;
    std::vector<GmContig> contigs;
};

bool gm_host_index_files_exist(const std::string& fa);
int gm_host_index_load(const std::string& fa, GmHostIndex& ix, std::string& err);
int gm_host_index_build(const std::string& fa, int where /* GM_BUILD_* */, int device_id, std::string& err);
// gm_sa_build.hip: suffix array / BWT / SA samples of the forward strand on the device
bool gm_device_available();
int gm_device_sa_build(const uint8_t* codes, uint64_t n, int device_id, uint32_t intv, std::vector<uint32_t>& plain, uint64_t& primary,
                       std::vector<uint64_t>& samples, int* rounds_out, std::string& err);

void gm_set_error(const std::string& s);
