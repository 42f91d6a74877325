// partition_check -- prints how rodent / bench_traversal divide n rows (or rays) among `world` GPUs (host/partition.h), one
// "rank begin end" line per GPU.  Test aid: tests/test_distributed.py compares it with rodent_amd/parallel.py.
#include <cstdlib>
#include <iostream>

#include "../partition.h"

int main(int argc, char** argv) {
    if (argc != 3) { std::cerr << "Usage: partition_check n world" << std::endl; return 1; }
    const int n = atoi(argv[1]), world = atoi(argv[2]);
    if (n < 0 || world < 1) { std::cerr << "Invalid arguments" << std::endl; return 1; }
    for (int r = 0; r < world; r++) { const rodent::Part p = rodent::split_range(n, r, world); std::cout << r << " " << p.begin << " " << p.end << "\n"; }
    return 0;
}
