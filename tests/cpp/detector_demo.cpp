/* Exercises the C++ adapter the way a ROS callback would: build a cloud, call filtered(), read the
 * four output clouds.  Built and run by tests/test_gpu_detector.py.
 *   usage: detector_demo rings cols scene seed out_labels.bin */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "detector.hpp"

int main(int argc, char** argv)
{
    if (argc != 6)
        return 2;
    const uint32_t rings = (uint32_t)atoi(argv[1]), cols = (uint32_t)atoi(argv[2]);
    const int scene = atoi(argv[3]);
    const uint64_t seed = (uint64_t)atoll(argv[4]);
    const uint32_t n = rings * cols;
    std::vector<float> x(n), y(n), z(n);
    if (urf_synth_cloud(rings, cols, scene, seed, x.data(), y.data(), z.data()) != URF_OK)
        return 3;
    urf::PointCloud cloud;
    cloud.header.frame_id = "left_os1/os1_lidar";
    cloud.points.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        cloud.points[i].x = x[i];
        cloud.points[i].y = y[i];
        cloud.points[i].z = z[i];
        cloud.points[i].intensity = (float)i;
    }
    try {
        urf::Detector det(0, n);
        urf_params p = det.params();
        p.min_X = p.min_Y = -200.f;
        p.max_X = p.max_Y = 200.f;
        det.setParams(p);
        det.setReferenceOrder(true);
        const bool published = det.filtered(cloud);
        std::printf("published %d road %zu curb %zu roi %zu road_probably %zu frame %s\n", (int)published,
                    det.road().points.size(), det.curb().points.size(), det.roi().points.size(),
                    det.road_probably().points.size(), det.road().header.frame_id.c_str());
        /* the clouds carry the original points: intensity is the input index */
        std::vector<uint8_t> lab(n, 0);
        for (const auto& q : det.roi().points) lab[(uint32_t)q.intensity] |= URF_FLAG_ROI;
        for (const auto& q : det.road().points) lab[(uint32_t)q.intensity] |= URF_LABEL_ROAD;
        for (const auto& q : det.curb().points) lab[(uint32_t)q.intensity] |= URF_LABEL_CURB;
        for (const auto& q : det.road_probably().points) lab[(uint32_t)q.intensity] |= URF_FLAG_RING10;
        /* the same sweep as a wire message with an Ouster-like field table (x y z intensity t ring):
         * must give the same clouds */
        {
            urf::PointCloud2 msg;
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
                float rec[8] = { 1.0f, z[i], 0.f, x[i], 0.f, y[i], 0.f, 0.f };
                std::memcpy(&msg.data[(size_t)i * 32], rec, 32);
            }
            urf::Detector det2(0, n);
            det2.setParams(p);
            const bool pub2 = det2.filtered(msg);
            std::printf("pc2 published %d same_labels %d\n", (int)pub2, (int)(det2.labels() == det.labels()));
        }
        FILE* f = std::fopen(argv[5], "wb");
        std::fwrite(lab.data(), 1, n, f);
        /* then the road cloud as a sequence of input indices, in published order */
        for (const auto& q : det.road().points) {
            const uint32_t id = (uint32_t)q.intensity;
            std::fwrite(&id, 4, 1, f);
        }
        std::fclose(f);
    } catch (const urf::Error& e) {
        std::fprintf(stderr, "urf error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
