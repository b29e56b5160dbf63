/* The multi-GPU pattern of SURVEY.md 8e from C++, WITH the collective the north-star names: one urf_ctx per GPU, scans
 * sharded by scan (scan s -> GPU s mod G, never split: ring table and beam march couple a scan's rings), no point data ever
 * crosses a link -- and at the end of the run one ncclAllReduce(sum) over six 64-bit counters and one ncclAllReduce(max) of
 * the elapsed time, through rccl.h directly (urban_road_filter_amd/sharding.py is the same thing through torch.distributed).
 * One process, one host thread and one RCCL communicator per device (ncclCommInitAll); on a box with one GPU G = 1 and the
 * collectives run over a communicator of one rank -- what a test box can prove (an 8-GPU node runs the same binary).
 *   usage: shard_demo out.bin n_gpus(0 = all) cloud.bin [cloud.bin ...]      cloud.bin: u32 n, float x[n], y[n], z[n]
 *   out.bin: per cloud n label bytes
 * Build: hipcc -std=c++17 -O2 shard_demo.cpp -I include -L urban_road_filter_amd -l:liburf_hip.so -lrccl -pthread */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "urf.h"

struct Cloud {
    uint32_t n = 0;
    std::vector<float> rec;   /* x y z pad per point (16-byte records) */
    std::vector<uint8_t> labels;
    urf_scan_info info{};
};

enum { N_COUNTERS = 6 };   /* scans, points_in, roi_points, road, curb, ok_scans (sharding.COUNTER_NAMES) */

int main(int argc, char** argv)
{
    if (argc < 4)
        return 2;
    std::vector<Cloud> clouds(argc - 3);
    uint32_t n_max = 0;
    for (int k = 3; k < argc; k++) {
        Cloud& c = clouds[k - 3];
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
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        std::fprintf(stderr, "no usable HIP device\n");
        return 1;
    }
    int G = std::atoi(argv[2]);
    if (G <= 0 || G > n_dev)
        G = n_dev;
    std::vector<int> devs(G);
    for (int g = 0; g < G; g++)
        devs[g] = g;
    std::vector<ncclComm_t> comms(G);
    if (ncclCommInitAll(comms.data(), G, devs.data()) != ncclSuccess) {   /* RCCL: one communicator per device, ranks 0..G-1 */
        std::fprintf(stderr, "ncclCommInitAll failed\n");
        return 1;
    }
    std::vector<int> rc_thread(G, 0);
    std::vector<std::vector<unsigned long long>> reduced(G, std::vector<unsigned long long>(N_COUNTERS + 1, 0));
    auto worker = [&](int g) {
        int rc = URF_OK;
        if (hipSetDevice(g) != hipSuccess) {
            rc_thread[g] = URF_ERR_NO_DEVICE;
            return;
        }
        urf_ctx* ctx = nullptr;
        rc = urf_create(&ctx, g, n_max, URF_MAX_IN_FLIGHT);
        urf_params p;
        urf_default_params(&p);
        p.min_X = p.min_Y = -200.f;
        p.max_X = p.max_Y = 200.f;
        if (rc == URF_OK)
            rc = urf_set_params(ctx, &p);
        unsigned long long local[N_COUNTERS] = { 0, 0, 0, 0, 0, 0 };
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t tickets[URF_MAX_IN_FLIGHT];
        int idx[URF_MAX_IN_FLIGHT], head = 0, count = 0;
        auto collect = [&]() {
            Cloud& c = clouds[idx[head]];
            const int r = urf_classify_pc2_wait(ctx, tickets[head], c.labels.data(), &c.info);
            if (r != URF_OK)
                rc = r;
            local[0]++;
            local[1] += c.n;
            local[2] += c.info.n_roi;
            local[3] += c.info.n_road;
            local[4] += c.info.n_curb;
            local[5] += c.info.status == URF_OK;
            head = (head + 1) % URF_MAX_IN_FLIGHT;
            count--;
        };
        for (int k = g; k < (int)clouds.size() && rc == URF_OK; k += G) {   /* this GPU's shard: scans g, g + G, ... */
            if (count == URF_MAX_IN_FLIGHT)
                collect();
            const int slot = (head + count) % URF_MAX_IN_FLIGHT;
            rc = urf_classify_pc2_async(ctx, (const uint8_t*)clouds[k].rec.data(), clouds[k].n, 16, 0, 4, 8, &tickets[slot]);
            idx[slot] = k;
            count++;
        }
        while (count && rc == URF_OK)
            collect();
        const unsigned long long ns = (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        /* the ONLY communication of the run: counters (sum) and elapsed time (max), 56 bytes over RCCL / xGMI.  Every rank
         * takes part even if its shard failed, so that no rank waits for ever. */
        unsigned long long* d = nullptr;
        hipStream_t st = nullptr;
        bool ok = hipMalloc((void**)&d, (N_COUNTERS + 1) * sizeof(unsigned long long)) == hipSuccess && hipStreamCreate(&st) == hipSuccess;
        unsigned long long h[N_COUNTERS + 1];
        std::memcpy(h, local, sizeof(local));
        h[N_COUNTERS] = ns;
        ok = ok && hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, st) == hipSuccess;
        ok = ok && ncclAllReduce(d, d, N_COUNTERS, ncclUint64, ncclSum, comms[g], st) == ncclSuccess;
        ok = ok && ncclAllReduce(d + N_COUNTERS, d + N_COUNTERS, 1, ncclUint64, ncclMax, comms[g], st) == ncclSuccess;
        ok = ok && hipMemcpyAsync(reduced[g].data(), d, sizeof(h), hipMemcpyDeviceToHost, st) == hipSuccess;
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        if (d)
            (void)hipFree(d);
        if (st)
            (void)hipStreamDestroy(st);
        if (ctx)
            urf_destroy(ctx);
        rc_thread[g] = rc != URF_OK ? rc : (ok ? URF_OK : URF_ERR_HIP);
    };
    std::vector<std::thread> threads;
    for (int g = 0; g < G; g++)
        threads.emplace_back(worker, g);
    for (auto& t : threads)
        t.join();
    for (int g = 0; g < G; g++)
        ncclCommDestroy(comms[g]);
    for (int g = 0; g < G; g++)
        if (rc_thread[g] != URF_OK) {
            std::fprintf(stderr, "rank %d: urf error %d\n", g, rc_thread[g]);
            return 1;
        }
    for (int g = 1; g < G; g++)   /* an all-reduce leaves the same totals on every rank */
        if (reduced[g] != reduced[0]) {
            std::fprintf(stderr, "ranks disagree about the reduced counters\n");
            return 1;
        }
    FILE* f = std::fopen(argv[1], "wb");
    for (const Cloud& c : clouds)
        std::fwrite(c.labels.data(), 1, c.n, f);
    std::fclose(f);
    const auto& r = reduced[0];
    std::printf("devices %d ranks %d scans %llu points_in %llu roi_points %llu road %llu curb %llu ok_scans %llu max_elapsed_ms %.3f scans_per_s %.1f\n", n_dev, G,
                r[0], r[1], r[2], r[3], r[4], r[5], r[6] * 1e-6, r[6] ? r[0] / (r[6] * 1e-9) : 0.0);
    return 0;
}
