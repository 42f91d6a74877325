// partition.h -- how a frame or a ray set is divided among the GPUs of a node: contiguous, disjoint, sizes
// differing by at most one.  Rows: 2160 rows on 8 GPUs = 270 each (the reference's tile arithmetic for frames is
// src/render/mapping_gpu.impala:384-420 / mapping_cpu.impala:200-237; seeds depend on absolute (sample, iter, x, y) only,
// renderer.impala:28-33, so any partition reproduces the frame).  Rays: contiguous ranges keep coherent rays coherent.
// The same arithmetic as rodent_amd/parallel.py (row_band, ray_range); tests/test_distributed.py compares the two.
#pragma once

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

} // namespace rodent
