// buffer_tool pack raw out.bin | unpack in.bin raw -- one buffer file of the reference's format (src/driver/buffer.h:
// [u32 size][u32 compressed size][LZ4 block]) from / to raw bytes; used by the codec tests against liblz4.
#include <cstdio>
#include <cstring>
#include <iostream>

#include "../buffer_io.h"

int main(int argc, char** argv) {
    if (argc != 4 || (strcmp(argv[1], "pack") && strcmp(argv[1], "unpack"))) {
        std::cerr << "usage: buffer_tool pack raw out.bin | unpack in.bin raw" << std::endl; return 1; }
    std::vector<uint8_t> data;
    if (!strcmp(argv[1], "pack")) {
        FILE* f = fopen(argv[2], "rb");
        if (!f) { std::cerr << "Cannot read '" << argv[2] << "'" << std::endl; return 1; }
        for (uint8_t buf[65536]; size_t n = fread(buf, 1, sizeof buf, f);) data.insert(data.end(), buf, buf + n);
        fclose(f);
        if (!rodent::write_buffer_file(argv[3], data)) { std::cerr << "Cannot write '" << argv[3] << "'" << std::endl; return 1; }
    } else {
        if (!rodent::read_buffer_file(argv[2], data)) { std::cerr << "Invalid buffer file '" << argv[2] << "'" << std::endl; return 1; }
        FILE* f = fopen(argv[3], "wb");
        if (!f || (data.size() && fwrite(data.data(), 1, data.size(), f) != data.size())) {
            std::cerr << "Cannot write '" << argv[3] << "'" << std::endl; return 1; }
        fclose(f);
    }
    return 0;
}
