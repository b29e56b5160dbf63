/* Exercises the C++ adapter the way a ROS callback would: build a cloud / a wire message, call filtered() (or
 * submit() / collect()), read the four output clouds -- and times the call, the unit the reference's callback is
 * (lidar_segmentation.cpp:95 -> :612-621: message in, four clouds out).  Links the PRODUCT library only.
 * Built and run by tests/test_gpu_detector.py and bench.py.
 *   usage: detector_demo cloud.bin out.bin [reps [default_roi]]      (default_roi: keep the reference's region of interest
 *                                                                      instead of widening x / y to +-200 m)
 *   cloud.bin: u32 n, float x[n], y[n], z[n], intensity[n]
 *   out.bin:   per run { u32 published; 4 x { u32 count; float xyzi[count][4] } }  (roi, road, curb, road_probably)
 *              runs: 0 PointCloud, reference order; 1 PointCloud2 with a permuted field table, input order;
 *                    2 PointCloud2 without an intensity field (point_step 23); 3 PointCloud2 in the Velodyne driver's layout;
 *                    4.. six sweeps through submit() / collect()
 *   stdout: "time <what> <median ms>" lines when reps > 0 */
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "detector.hpp"

static void dump(FILE* f, bool published, const urf::Detector& det)
{
    const uint32_t pub = published ? 1u : 0u;
    std::fwrite(&pub, 4, 1, f);
    for (const urf::PointCloud* c : { &det.roi(), &det.road(), &det.curb(), &det.road_probably() }) {
        const uint32_t n = (uint32_t)c->points.size();
        std::fwrite(&n, 4, 1, f);
        for (const auto& q : c->points) {
            const float v[4] = { q.x, q.y, q.z, q.intensity };
            std::fwrite(v, 4, 4, f);
        }
    }
}

template <class F>
static double median_ms(int reps, F&& fn)
{
    std::vector<double> t;
    for (int k = 0; k < reps + 3; k++) {
        const auto t0 = std::chrono::steady_clock::now();
        fn();
        const auto t1 = std::chrono::steady_clock::now();
        if (k >= 3)
            t.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv)
{
    if (argc < 3)
        return 2;
    const int reps = argc > 3 ? atoi(argv[3]) : 0;
    FILE* fi = std::fopen(argv[1], "rb");
    if (!fi)
        return 3;
    uint32_t n = 0;
    if (std::fread(&n, 4, 1, fi) != 1)
        return 3;
    std::vector<float> x(n), y(n), z(n), in(n);
    if (std::fread(x.data(), 4, n, fi) != n || std::fread(y.data(), 4, n, fi) != n || std::fread(z.data(), 4, n, fi) != n ||
        std::fread(in.data(), 4, n, fi) != n)
        return 3;
    std::fclose(fi);
    urf::PointCloud cloud;
    cloud.header.frame_id = "left_os1/os1_lidar";
    cloud.points.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        cloud.points[i].x = x[i];
        cloud.points[i].y = y[i];
        cloud.points[i].z = z[i];
        cloud.points[i].intensity = in[i];
    }
    /* the same sweep as a wire message with an Ouster-like field table (intensity z t x ring y) */
    urf::PointCloud2 msg;
    {
        msg.header = cloud.header;
        msg.width = n;
        msg.point_step = 32;
        const char* names[6] = { "intensity", "z", "t", "x", "ring", "y" };
        const uint32_t offs[6] = { 0, 4, 8, 12, 16, 20 };
        const uint8_t types[6] = { urf::PointField::FLOAT32, urf::PointField::FLOAT32, urf::PointField::UINT32,
                                   urf::PointField::FLOAT32, urf::PointField::UINT16, urf::PointField::FLOAT32 };
        for (int k = 0; k < 6; k++) {
            urf::PointField pf;
            pf.name = names[k];
            pf.offset = offs[k];
            pf.datatype = types[k];
            msg.fields.push_back(pf);
        }
        msg.data.resize((size_t)n * 32);
        for (uint32_t i = 0; i < n; i++) {
            float rec[8] = { in[i], z[i], 0.f, x[i], 0.f, y[i], 0.f, 0.f };
            std::memcpy(&msg.data[(size_t)i * 32], rec, 32);
        }
    }
    /* ... as a Velodyne driver publishes it: x y z intensity (FLOAT32 at 0 / 4 / 8 / 12), ring (UINT16 at 16), time (FLOAT32 at 18), 22 bytes */
    urf::PointCloud2 velo;
    {
        velo.header = cloud.header;
        velo.width = n;
        velo.point_step = 22;
        const char* names[6] = { "x", "y", "z", "intensity", "ring", "time" };
        const uint32_t offs[6] = { 0, 4, 8, 12, 16, 18 };
        const uint8_t types[6] = { urf::PointField::FLOAT32, urf::PointField::FLOAT32, urf::PointField::FLOAT32,
                                   urf::PointField::FLOAT32, urf::PointField::UINT16, urf::PointField::FLOAT32 };
        for (int k = 0; k < 6; k++) {
            urf::PointField pf;
            pf.name = names[k];
            pf.offset = offs[k];
            pf.datatype = types[k];
            velo.fields.push_back(pf);
        }
        velo.data.assign((size_t)n * 22, 0);
        for (uint32_t i = 0; i < n; i++) {
            const float rec[4] = { x[i], y[i], z[i], in[i] };
            std::memcpy(&velo.data[(size_t)i * 22], rec, 16);
        }
    }
    /* ... and with unaligned 23-byte records that carry no intensity (x at 3, y at 11, z at 17) */
    urf::PointCloud2 bare;
    {
        bare.header = cloud.header;
        bare.width = n;
        bare.point_step = 23;
        const char* names[3] = { "x", "y", "z" };
        const uint32_t offs[3] = { 3, 11, 17 };
        for (int k = 0; k < 3; k++) {
            urf::PointField pf;
            pf.name = names[k];
            pf.offset = offs[k];
            pf.datatype = urf::PointField::FLOAT32;
            bare.fields.push_back(pf);
        }
        bare.data.assign((size_t)n * 23, 0xa5);
        for (uint32_t i = 0; i < n; i++) {
            std::memcpy(&bare.data[(size_t)i * 23 + 3], &x[i], 4);
            std::memcpy(&bare.data[(size_t)i * 23 + 11], &y[i], 4);
            std::memcpy(&bare.data[(size_t)i * 23 + 17], &z[i], 4);
        }
    }
    try {
        urf::Detector det(0, n);
        urf_params p = det.params();
        if (!(argc > 4 && std::strcmp(argv[4], "default_roi") == 0)) {
            p.min_X = p.min_Y = -200.f;
            p.max_X = p.max_Y = 200.f;
        }
        det.setParams(p);
        FILE* f = std::fopen(argv[2], "wb");
        det.setReferenceOrder(true);
        bool published = det.filtered(cloud);
        std::printf("published %d road %zu curb %zu roi %zu road_probably %zu frame %s\n", (int)published,
                    det.road().points.size(), det.curb().points.size(), det.roi().points.size(),
                    det.road_probably().points.size(), det.road().header.frame_id.c_str());
        dump(f, published, det);
        std::vector<uint8_t> lab0(det.labels(), det.labels() + det.n_labels());
        det.setReferenceOrder(false);
        published = det.filtered(msg);
        std::printf("pc2 published %d same_labels %d\n", (int)published,
                    (int)(det.n_labels() == lab0.size() && std::memcmp(det.labels(), lab0.data(), lab0.size()) == 0));
        dump(f, published, det);
        published = det.filtered(bare);
        std::printf("bare published %d same_labels %d\n", (int)published,
                    (int)(det.n_labels() == lab0.size() && std::memcmp(det.labels(), lab0.data(), lab0.size()) == 0));
        dump(f, published, det);
        published = det.filtered(velo);
        std::printf("velodyne published %d same_labels %d\n", (int)published,
                    (int)(det.n_labels() == lab0.size() && std::memcmp(det.labels(), lab0.data(), lab0.size()) == 0));
        dump(f, published, det);
        /* six sweeps, four in flight (the subscriber callback submits, the publisher collects) */
        {
            uint32_t tickets[6];
            int head = 0, count = 0, same = 0, done = 0;
            for (int k = 0; k < 6; k++) {
                if (count == URF_MAX_IN_FLIGHT) {
                    published = det.collect(tickets[head++]);
                    count--;
                    same += det.n_labels() == lab0.size() && std::memcmp(det.labels(), lab0.data(), lab0.size()) == 0;
                    dump(f, published, det);
                    done++;
                }
                tickets[k] = (k & 1) ? det.submit(msg) : det.submit(cloud);
                count++;
            }
            while (count--) {
                published = det.collect(tickets[head++]);
                same += det.n_labels() == lab0.size() && std::memcmp(det.labels(), lab0.data(), lab0.size()) == 0;
                dump(f, published, det);
                done++;
            }
            std::printf("pipelined sweeps %d same_labels %d\n", done, same);
        }
        std::fclose(f);
        if (reps > 0) {
            det.setReferenceOrder(false);
            std::printf("time pointcloud_input_order %.4f\n", median_ms(reps, [&] { det.filtered(cloud); }));
            std::printf("time pointcloud2_permuted_fields %.4f\n", median_ms(reps, [&] { det.filtered(msg); }));
            std::printf("time pointcloud2_velodyne_layout %.4f\n", median_ms(reps, [&] { det.filtered(velo); }));
            std::printf("time pointcloud2_xyz_only_step23 %.4f\n", median_ms(reps, [&] { det.filtered(bare); }));
            det.setReferenceOrder(true);
            std::printf("time pointcloud_reference_order %.4f\n", median_ms(reps, [&] { det.filtered(cloud); }));
            det.setReferenceOrder(false);
            det.enableRoadMarker(true);
            std::printf("time pointcloud_input_order_with_marker %.4f\n", median_ms(reps, [&] { det.filtered(cloud); }));
            det.enableRoadMarker(false);
            /* throughput with four sweeps in flight: submit until the slots are full, collect the oldest */
            {
                const int total = 8 * reps;
                uint32_t ring[URF_MAX_IN_FLIGHT];
                int head = 0, count = 0;
                const auto t0 = std::chrono::steady_clock::now();
                for (int k = 0; k < total; k++) {
                    if (count == URF_MAX_IN_FLIGHT) {
                        det.collect(ring[head]);
                        head = (head + 1) % URF_MAX_IN_FLIGHT;
                        count--;
                    }
                    ring[(head + count) % URF_MAX_IN_FLIGHT] = det.submit(cloud);
                    count++;
                }
                while (count--) {
                    det.collect(ring[head]);
                    head = (head + 1) % URF_MAX_IN_FLIGHT;
                }
                const auto t1 = std::chrono::steady_clock::now();
                std::printf("time pipelined_per_sweep %.4f\n", std::chrono::duration<double, std::milli>(t1 - t0).count() / total);
            }
        }
    } catch (const urf::Error& e) {
        std::fprintf(stderr, "urf error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
