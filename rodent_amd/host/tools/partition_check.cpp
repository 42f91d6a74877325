// partition_check -- prints how rodent / bench_traversal divide n rows (or rays) among `world` GPUs (host/partition.h), one
// "rank begin end" line per GPU.  Test aid: tests/test_distributed.py compares it with rodent_amd/parallel.py.
#include <cstdlib>
#include <iostream>
#include <string>

#include "../partition.h"

int main(int argc, char** argv) {
    // partition_check transport ranks devices shared injected [init error]: the gather's transport decision (host/partition.h
    // rccl_unused_reason)
    if (argc >= 6 && std::string(argv[1]) == "transport") {
        const std::string why = rodent::rccl_unused_reason(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]) != 0, atoi(argv[5]) != 0,
            argc > 6 ? argv[6] : "");
        std::cout << (why.empty() ? "rccl" : "peer copies: " + why) << "\n";
        return 0;
    }
    if (argc != 3 && argc != 4) { std::cerr << "Usage: partition_check n world [tile_rows]" << std::endl; return 1; }
    const int n = atoi(argv[1]), world = atoi(argv[2]), tile_rows = argc == 4 ? atoi(argv[3]) : 0;
    if (n < 0 || world < 1 || tile_rows < 0) { std::cerr << "Invalid arguments" << std::endl; return 1; }
    if (tile_rows > 0) {                                   // interleaved row tiles: one "rank begin end" line per tile
        for (int r = 0; r < world; r++) rodent::for_each_tile(n, r, world, tile_rows,
            [&](rodent::Part p) { std::cout << r << " " << p.begin << " " << p.end << "\n"; });
        return 0;
    }
    for (int r = 0; r < world; r++) { const rodent::Part p = rodent::split_range(n, r, world);
        std::cout << r << " " << p.begin << " " << p.end << "\n"; }
    return 0;
}
