// partition.h -- how a frame or a ray set is divided among the GPUs of a node: contiguous, disjoint, sizes
// differing by at most one.  Rows: 2160 rows on 8 GPUs = 270 each (the reference's tile arithmetic for frames is
// src/render/mapping_gpu.impala:384-420 / mapping_cpu.impala:200-237; seeds depend on absolute (sample, iter, x, y) only,
// renderer.impala:28-33, so any partition reproduces the frame).  Rays: contiguous ranges keep coherent rays coherent.
// The same arithmetic as rodent_amd/parallel.py (row_band, ray_range); tests/test_distributed.py compares the two.
#pragma once
#include <string>

namespace rodent {

struct Part {
    int begin, end;
    int size() const { return end - begin; }
};

// part `rank` of `world` of the index range [0, n)
inline Part split_range(int n, int rank, int world) {
    const int base = n / world, extra = n % world;
    const int begin = rank * base + (rank < extra ? rank : extra);
    return Part{begin, begin + base + (rank < extra ? 1 : 0)};
}

// Interleaved row tiles (SURVEY 8e "interleaved 16-row tiles for load balance"; the reference deals ~1024-sample tiles dynamically,
// src/render/mapping_gpu.impala:374-420): rank r owns the tiles r, r + world, ... of tile_rows rows each (the film's last tile may be
// shorter).  Contiguous bands of the atrium frame differ by 27 % in cost (profiles/r04_band_costs.txt); shares of interleaved tiles by 1 %.
// The same arithmetic as rodent_amd/parallel.py row_tiles.
constexpr int kTileRows = 16;
template <typename F>   // f(Part rows) for every tile of `rank`, top to bottom
inline void for_each_tile(int height, int rank, int world, int tile_rows, F f) {
    for (int t = rank; t * tile_rows < height; t += world) f(Part{t * tile_rows,
        (t + 1) * tile_rows < height ? (t + 1) * tile_rows : height});
}
inline int tile_rows_of_rank(int height, int rank, int world, int tile_rows) {
    int rows = 0;
    for_each_tile(height, rank, world, tile_rows, [&](Part p) { rows += p.size(); });
    return rows;
}

// Which transport the ONE gather of a K-GPU run uses (host/multi_gpu.h DeviceGroup::init): RCCL when it comes up, otherwise one peer copy
// per piece -- a run on several GPUs must not fail for want of a collective library, and must say what it did.  Pure decision logic (no HIP
// / RCCL here) so that the CPU suite can test the fallback branch with an injected failure: returns the reason RCCL is NOT used, empty = it
// is.
//   ranks > devices (RODENT_SHARE_GPUS: ranks share devices; RCCL cannot place two ranks on one device), an injected failure
//   (RODENT_FORCE_RCCL_INIT_FAILURE), or ncclCommInitAll's own error string.
inline std::string rccl_unused_reason(int ranks, int devices, bool shared, bool injected_failure, const std::string& init_error) {
    if (ranks <= 1) return "";
    if (shared && devices < ranks) return "RODENT_SHARE_GPUS: " + std::to_string(ranks) + " ranks on " + std::to_string(devices)
        + " device(s)";
    if (injected_failure) return "RODENT_FORCE_RCCL_INIT_FAILURE";
    if (!init_error.empty()) return "ncclCommInitAll: " + init_error;
    return "";
}

} // namespace rodent
