/* One context per GPU from C++: what `bench.py --gpus N` does with one process per GPU, done here with one host thread per
 * context inside ONE process (include/urf.h: "a context is not thread-safe, any number of contexts may coexist").  Context i
 * lives on device i % (number of devices) -- on a 1-GPU box both share device 0 --, each thread classifies its own sweeps
 * through the asynchronous path, and the only thing the threads exchange is what the multi-GPU benchmark exchanges: counters.
 *   usage: two_contexts_demo out.bin cloud.bin [cloud.bin ...]      cloud.bin: u32 n, float x[n], y[n], z[n]
 *   out.bin: per cloud n label bytes (cloud k is classified by context k % 2) */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "urf.h"

struct Cloud {
    uint32_t n = 0;
    std::vector<float> rec;   /* x y z pad per point (16-byte records) */
    std::vector<uint8_t> labels;
    urf_scan_info info{};
};

int main(int argc, char** argv)
{
    if (argc < 3)
        return 2;
    std::vector<Cloud> clouds(argc - 2);
    uint32_t n_max = 0;
    for (int k = 2; k < argc; k++) {
        Cloud& c = clouds[k - 2];
        FILE* f = std::fopen(argv[k], "rb");
        if (!f || std::fread(&c.n, 4, 1, f) != 1)
            return 3;
        std::vector<float> x(c.n), y(c.n), z(c.n);
        if (std::fread(x.data(), 4, c.n, f) != c.n || std::fread(y.data(), 4, c.n, f) != c.n || std::fread(z.data(), 4, c.n, f) != c.n)
            return 3;
        std::fclose(f);
        c.rec.assign((size_t)c.n * 4, 0.f);
        for (uint32_t i = 0; i < c.n; i++) {
            c.rec[4 * (size_t)i] = x[i];
            c.rec[4 * (size_t)i + 1] = y[i];
            c.rec[4 * (size_t)i + 2] = z[i];
        }
        c.labels.assign(c.n, 0xEE);
        n_max = c.n > n_max ? c.n : n_max;
    }
    /* number of devices: contexts are created on device 0, 1, ... until urf_create refuses one */
    int n_dev = 0;
    {
        urf_ctx* probe = nullptr;
        while (n_dev < 16 && urf_create(&probe, n_dev, 1024, 1) == URF_OK) {
            urf_destroy(probe);
            n_dev++;
        }
    }
    if (n_dev == 0) {
        std::fprintf(stderr, "no device\n");
        return 1;
    }
    const int n_ctx = 2;
    int rc_thread[n_ctx] = { 0, 0 };
    uint64_t counters[n_ctx][3] = { { 0, 0, 0 }, { 0, 0, 0 } };   /* sweeps, road points, curb points: all that is ever exchanged */
    auto worker = [&](int w) {
        urf_ctx* ctx = nullptr;
        int rc = urf_create(&ctx, w % n_dev, n_max, URF_MAX_IN_FLIGHT);
        if (rc != URF_OK) {
            rc_thread[w] = rc;
            return;
        }
        urf_params p;
        urf_default_params(&p);
        p.min_X = p.min_Y = -200.f;
        p.max_X = p.max_Y = 200.f;
        if (w == 1)
            p.curbHeight = 0.06f;   /* the contexts do not share parameters either */
        rc = urf_set_params(ctx, &p);
        uint32_t tickets[URF_MAX_IN_FLIGHT];
        int idx[URF_MAX_IN_FLIGHT], head = 0, count = 0;
        auto collect = [&]() {
            Cloud& c = clouds[idx[head]];
            const int r = urf_classify_pc2_wait(ctx, tickets[head], c.labels.data(), &c.info);
            if (r != URF_OK)
                rc = r;
            counters[w][0]++;
            counters[w][1] += c.info.n_road;
            counters[w][2] += c.info.n_curb;
            head = (head + 1) % URF_MAX_IN_FLIGHT;
            count--;
        };
        for (int k = w; k < (int)clouds.size() && rc == URF_OK; k += n_ctx) {
            if (count == URF_MAX_IN_FLIGHT)
                collect();
            const int slot = (head + count) % URF_MAX_IN_FLIGHT;
            rc = urf_classify_pc2_async(ctx, (const uint8_t*)clouds[k].rec.data(), clouds[k].n, 16, 0, 4, 8, &tickets[slot]);
            idx[slot] = k;
            count++;
        }
        while (count && rc == URF_OK)
            collect();
        rc_thread[w] = rc;
        urf_destroy(ctx);
    };
    std::thread t0(worker, 0), t1(worker, 1);
    t0.join();
    t1.join();
    if (rc_thread[0] != URF_OK || rc_thread[1] != URF_OK) {
        std::fprintf(stderr, "urf error %d / %d\n", rc_thread[0], rc_thread[1]);
        return 1;
    }
    FILE* f = std::fopen(argv[1], "wb");
    for (const Cloud& c : clouds)
        std::fwrite(c.labels.data(), 1, c.n, f);
    std::fclose(f);
    std::printf("devices %d contexts %d sweeps %llu + %llu road %llu curb %llu\n", n_dev, n_ctx, (unsigned long long)counters[0][0],
                (unsigned long long)counters[1][0], (unsigned long long)(counters[0][1] + counters[1][1]),
                (unsigned long long)(counters[0][2] + counters[1][2]));
    return 0;
}
