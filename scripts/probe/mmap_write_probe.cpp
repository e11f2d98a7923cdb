// What a snapshot SAVE can reach on one file: 16 threads, 8 GB from a memory buffer into ONE fresh file on tmpfs --
// pwrite (serialised by the inode lock) against memcpy into a MAP_SHARED mapping (with / without MADV_POPULATE_WRITE).
//   g++ -O2 -pthread -o mmap_write_probe mmap_write_probe.cpp && ./mmap_write_probe [GB] [DIR] [THREADS]
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
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
int main(int argc, char **argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 8.0;
    const std::string dir = argc > 2 ? argv[2] : "/dev/shm";
    const unsigned nt = argc > 3 ? atoi(argv[3]) : 16;
    const size_t chunk = 256ull << 20, per = (size_t)(gb * 1e9 / nt) / (4 << 20) * (4 << 20), total = per * nt;
    std::vector<char> src(chunk);
    for (size_t i = 0; i < chunk; i++) src[i] = (char)(i * 2654435761u >> 13);
    const std::string path = dir + "/mmap_probe_" + std::to_string(getpid());
    for (int mode = 0; mode < 4; mode++) {
        const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) { perror("open"); return 1; }
        char *map = nullptr;
        if (mode) {
            if (ftruncate(fd, (off_t)total) != 0) { perror("ftruncate"); return 1; }
            map = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (map == MAP_FAILED) { perror("mmap"); return 1; }
        }
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                // chunks of 256 MB as the library's pinned buffers are; each thread its 1/nt of every chunk (mode 3) or its own range
                const size_t a = (size_t)t * per;
                if (mode == 2 && madvise(map + a, per, MADV_POPULATE_WRITE) != 0) perror("madvise");
                for (size_t o = 0; o < per; o += 4 << 20) {
                    const size_t n = std::min<size_t>(4 << 20, per - o);
                    if (mode == 0) { if (pwrite(fd, src.data() + (o % chunk), n, (off_t)(a + o)) != (ssize_t)n) perror("pwrite"); }
                    else {
                        if (mode == 3 && madvise(map + a + o, n, MADV_POPULATE_WRITE) != 0) perror("madvise");
                        memcpy(map + a + o, src.data() + (o % chunk), n);
                    }
                }
            });
        for (auto &x : th) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const char *names[] = {"pwrite, one fresh file", "mmap + memcpy (page faults)", "mmap + MADV_POPULATE_WRITE per thread range + memcpy", "mmap + MADV_POPULATE_WRITE per 4 MB + memcpy"};
        printf("%-58s %6.2f GB/s\n", names[mode], total / dt / 1e9);
        fflush(stdout);
        if (map) munmap(map, total);
        close(fd);
        unlink(path.c_str());
    }
    return 0;
}
