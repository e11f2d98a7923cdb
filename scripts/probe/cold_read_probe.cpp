// Why is the FIRST read of a freshly written tmpfs file 2-4x slower than the second (snapshot load 20 vs 44 GB/s, BerkeleyDB scan
// 10 vs 70 GB/s)?  The same 8 GB file, written then read for the first time by 16 threads: pread into a buffer vs memcpy out of a
// MAP_SHARED mapping (with / without MADV_POPULATE_READ), then each once more (warm).
//   g++ -O2 -pthread -o cold_read_probe cold_read_probe.cpp && ./cold_read_probe [GB] [DIR] [THREADS]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 8.0;
    const std::string dir = argc > 2 ? argv[2] : "/dev/shm";
    const unsigned nt = argc > 3 ? atoi(argv[3]) : 16;
    const size_t blk = 4 << 20, per = (size_t)(gb * 1e9 / nt) / blk * blk, total = per * nt;
    std::vector<std::vector<char>> bufs(nt, std::vector<char>(blk));
    auto write_files = [&](const std::string &base) {          // striped like the snapshot: nt part files, one writer each (fast)
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                const int fd = open((base + std::to_string(t)).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
                memset(bufs[t].data(), (int)t + 1, blk);
                for (size_t o = 0; o < per; o += blk) if (pwrite(fd, bufs[t].data(), blk, (off_t)o) != (ssize_t)blk) perror("pwrite");
                close(fd);
            });
        for (auto &x : th) x.join();
    };
    const std::string base = dir + "/cold_probe_" + std::to_string(getpid()) + "_";
    for (int mode = 0; mode < 3; mode++) {
        write_files(base);
        for (int pass = 0; pass < 2; pass++) {
            const double t0 = now();
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++)
                th.emplace_back([&, t]() {
                    const int fd = open((base + std::to_string(t)).c_str(), O_RDONLY);
                    if (mode == 0) {
                        for (size_t o = 0; o < per; o += blk) if (pread(fd, bufs[t].data(), blk, (off_t)o) != (ssize_t)blk) perror("pread");
                    } else {
                        char *map = (char *)mmap(nullptr, per, PROT_READ, MAP_SHARED, fd, 0);
                        if (map == MAP_FAILED) { perror("mmap"); return; }
                        for (size_t o = 0; o < per; o += blk) {
                            if (mode == 2 && madvise(map + o, blk, MADV_POPULATE_READ) != 0) perror("madvise");
                            memcpy(bufs[t].data(), map + o, blk);
                        }
                        munmap(map, per);
                    }
                    close(fd);
                });
            for (auto &x : th) x.join();
            const char *names[] = {"pread", "mmap + memcpy (page faults)", "mmap + MADV_POPULATE_READ per 4 MB + memcpy"};
            printf("%-46s %s read %7.2f GB/s\n", names[mode], pass ? "second" : "FIRST ", total / (now() - t0) / 1e9);
            fflush(stdout);
        }
        for (unsigned t = 0; t < nt; t++) unlink((base + std::to_string(t)).c_str());
    }
    return 0;
}
