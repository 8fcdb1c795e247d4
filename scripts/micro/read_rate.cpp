// Measurement aid: how fast a process gets a file of the page cache (tmpfs) into a buffer of its own -- pread against mmap + memcpy, by thread
// count, into plain and into touched-before memory.    g++ -O2 -pthread -o read_rate read_rate.cpp && ./read_rate /dev/shm/file
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/dev/shm/read_rate.bin";
    int fd = open(path, O_RDONLY);
    if (fd < 0) {   // make a 1 GiB file
        fd = open(path, O_CREAT | O_RDWR, 0600);
        std::vector<char> blk(1 << 24, 'A');
        for (int i = 0; i < 64; i++) if (write(fd, blk.data(), blk.size()) < 0) return 1;
        close(fd);
        fd = open(path, O_RDONLY);
    }
    struct stat st;
    fstat(fd, &st);
    const size_t n = (size_t)st.st_size;
    char* buf = (char*)aligned_alloc(4096, n);
    memset(buf, 1, n);                                        // touched before: what a pinned buffer is
    const char* map = (const char*)mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
    for (int mode = 0; mode < 2; mode++)
        for (int threads : {1, 2, 4, 8, 16}) {
            for (int rep = 0; rep < 2; rep++) {
                const double t0 = now();
                std::vector<std::thread> th;
                const size_t part = ((n + threads - 1) / threads + 4095) & ~(size_t)4095;
                for (int t = 0; t < threads; t++)
                    th.emplace_back([=] {
                        const size_t lo = (size_t)t * part, hi = lo + part < n ? lo + part : n;
                        if (lo >= hi) return;
                        if (mode == 0) { size_t got = 0; while (got < hi - lo) { ssize_t r = pread(fd, buf + lo + got, hi - lo - got, (off_t)(lo + got)); if (r <= 0) break; got += (size_t)r; } }
                        else memcpy(buf + lo, map + lo, hi - lo);
                    });
                for (auto& x : th) x.join();
                const double dt = now() - t0;
                if (rep) printf("%s, %2d threads: %.1f ms = %.1f GB/s\n", mode ? "mmap + memcpy" : "pread        ", threads, 1e3 * dt, n / dt / 1e9);
            }
        }
    return 0;
}
