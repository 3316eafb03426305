// How fast does ONE file in a directory take bytes from memory?  g++ -O2 -std=c++17 -pthread exp/tmpfs_write_bench.cpp -o /tmp/twb; /tmp/twb /dev/shm/x 6 8
// modes: write (one thread, 64 MB calls), pwrite by W threads, mmap + memcpy by W threads (with and without MADV_POPULATE_WRITE)
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const char *path = argv[1];
    const size_t gb = argc > 2 ? atoi(argv[2]) : 4, chunk = 64u << 20, total = gb << 30;
    const int max_w = argc > 3 ? atoi(argv[3]) : 8;
    std::vector<char> src(chunk);
    for (size_t i = 0; i < chunk; ++i) src[i] = (char)(i * 131);
    auto report = [&](const char *what, int w, double t) { printf("%-28s W=%2d  %.2f GB/s\n", what, w, (double)total / t / 1e9); fflush(stdout); };
    {
        unlink(path);
        int fd = open(path, O_CREAT | O_WRONLY | O_TRUNC, 0644);
        double t0 = now();
        for (size_t off = 0; off < total; off += chunk)
            if (write(fd, src.data(), chunk) != (ssize_t)chunk) return 1;
        report("write", 1, now() - t0);
        close(fd);
    }
    for (int w = 2; w <= max_w; w *= 2) {
        unlink(path);
        int fd = open(path, O_CREAT | O_WRONLY | O_TRUNC, 0644);
        double t0 = now();
        for (size_t off = 0; off < total; off += chunk) {
            std::vector<std::thread> th;
            for (int k = 0; k < w; ++k)
                th.emplace_back([&, k] {
                    const size_t a = chunk / w * k, n = chunk / w;
                    if (pwrite(fd, src.data() + a, n, off + a) != (ssize_t)n) abort();
                });
            for (auto &t : th) t.join();
        }
        report("pwrite, slices of a chunk", w, now() - t0);
        close(fd);
    }
    for (int populate = 0; populate < 2; ++populate)
        for (int w = 1; w <= max_w; w *= 2) {
            unlink(path);
            int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
            double t0 = now();
            for (size_t off = 0; off < total; off += chunk) {
                if (ftruncate(fd, off + chunk)) return 1;
                std::vector<std::thread> th;
                for (int k = 0; k < w; ++k)
                    th.emplace_back([&, k] {
                        const size_t a = chunk / w * k, n = chunk / w;
                        char *m = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, off + a);
                        if (m == MAP_FAILED) abort();
                        if (populate) madvise(m, n, MADV_POPULATE_WRITE);
                        memcpy(m, src.data() + a, n);
                        munmap(m, n);
                    });
                for (auto &t : th) t.join();
            }
            report(populate ? "mmap + populate + memcpy" : "mmap + memcpy", w, now() - t0);
            close(fd);
        }
    unlink(path);
    return 0;
}
