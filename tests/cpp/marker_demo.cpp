/* Runs a short SEQUENCE of sweeps through one urf::Detector with the road_marker output enabled
 * (the marker builder keeps state between sweeps, like the reference) and dumps the MarkerArrays.
 *   usage: marker_demo simple_poly_allow poly_z_avg_allow out.bin  cloud.bin [cloud.bin ...]
 *   cloud.bin: u32 n, float x[n], y[n], z[n]   (links the product library only: the sweeps come from the test)
 *   out: per sweep { u32 published, u32 n_markers, n_markers x { i32 id, action, type; f32 rgba[4];
 *        u32 n_points; f64 xyz[n_points][3] } } */
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "detector.hpp"

int main(int argc, char** argv)
{
    if (argc < 5)
        return 2;
    urf_marker_params mp;
    urf_default_marker_params(&mp);
    mp.simple_poly_allow = atoi(argv[1]);
    mp.poly_z_avg_allow = atoi(argv[2]);
    FILE* f = std::fopen(argv[3], "wb");
    const uint32_t n_max = 64 * 2048;
    try {
        urf::Detector det(0, n_max);
        urf_params p = det.params();
        p.min_X = p.min_Y = -200.f;
        p.max_X = p.max_Y = 200.f;
        det.setParams(p);
        det.enableRoadMarker(true);
        det.setMarkerParams(mp);
        for (int k = 4; k < argc; k++) {
            FILE* fi = std::fopen(argv[k], "rb");
            uint32_t n = 0;
            if (!fi || std::fread(&n, 4, 1, fi) != 1 || n > n_max)
                return 3;
            std::vector<float> x(n), y(n), z(n);
            if (std::fread(x.data(), 4, n, fi) != n || std::fread(y.data(), 4, n, fi) != n || std::fread(z.data(), 4, n, fi) != n)
                return 3;
            std::fclose(fi);
            urf::PointCloud cloud;
            cloud.points.resize(n);
            for (uint32_t i = 0; i < n; i++) {
                cloud.points[i].x = x[i];
                cloud.points[i].y = y[i];
                cloud.points[i].z = z[i];
            }
            det.filtered(cloud);
            const urf::MarkerArray* ma = det.road_marker();
            const uint32_t pub = ma ? 1u : 0u, nm = ma ? (uint32_t)ma->markers.size() : 0u;
            std::fwrite(&pub, 4, 1, f);
            std::fwrite(&nm, 4, 1, f);
            for (uint32_t m = 0; m < nm; m++) {
                const urf::Marker& mk = ma->markers[m];
                const int32_t ia[3] = { mk.id, mk.action, mk.type };
                const uint32_t np = (uint32_t)mk.points.size();
                std::fwrite(ia, 4, 3, f);
                std::fwrite(mk.color.data(), 4, 4, f);
                std::fwrite(&np, 4, 1, f);
                for (const auto& q : mk.points)
                    std::fwrite(q.data(), 8, 3, f);
            }
        }
    } catch (const urf::Error& e) {
        std::fprintf(stderr, "urf error %d: %s\n", e.code, e.what());
        return 1;
    }
    std::fclose(f);
    return 0;
}
