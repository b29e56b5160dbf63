/*
 * ref_harness.cpp -- "oracle A": drives the reference's OWN, UNMODIFIED
 * hot-path sources (compiled from /root/reference by oracle/Makefile against
 * the stand-in headers in oracle/shim/) and dumps per-point labels.
 *
 * TEST INFRASTRUCTURE ONLY.  Used to (1) pin oracle B (oracle/urf_oracle.c),
 * (2) generate tests/golden/, (3) time the reference's CPU path for
 * bench.py's cpu_baseline leg ("kind": "reference").  Never part of the
 * product path.
 *
 * It is a stand-alone executable, not a library, because
 *   - the reference keeps its state in globals (one Detector per process),
 *   - it needs a process-wide guard allocator: blind_spots.cpp:216/233/255/273
 *     read array3D[k][-1].alpha before testing j >= 0, which faults when the
 *     ring vector is an mmap'ed chunk; operator new below
 *     puts 64 zero bytes in front of every allocation,
 *   - lidar_segmentation.cpp:298 is a stack VLA of piece*16 bytes, so the
 *     callback runs on a thread with a large stack.
 *
 * Usage:  urf_ref <in.bin> <out.bin>
 *   in : "URFREFIN" u32 n_scans u32 n_points u32 repeat u32 flags
 *        urf_params (sizeof) [urf_marker_params if flags & 1] then n_scans x { x[n] y[n] z[n] } float32
 *   out: "URFREFOU" u32 n_scans u32 n_points f64 ms_per_scan_steady f64 ms_first
 *        then n_scans x { urf_scan_info, labels[n], road[n_road], curb[n_curb], road_probably[n_ring10] }
 *        (u32 input indices in the order the reference published them)
 *        and, if flags & 1, the road_marker MarkerArray of the scan: u32 published, u32 n_markers,
 *        n_markers x { i32 id, action, type; f32 r, g, b, a; u32 n_points; f64 xyz[n_points][3] }
 *        (the scans are run in sequence by ONE Detector, so ghostcount and the member linestring
 *        carry over from scan to scan exactly as in the node)
 * Points are identified by writing the input index into `intensity`
 * (exact in float up to 2^24 points).  `repeat` > 1 re-runs the whole set for
 * timing; the first call of the process is excluded from the steady figure.
 * The RING bit (0x08) of the label byte and info.n_rings/n_ring_pts are not
 * observable from the reference's outputs and stay 0.
 */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <pthread.h>

#include "urban_road_filter/data_structures.hpp"
#include "urf.h"

/* ---- guard allocator ---------------------------------------------------- */
static const size_t GUARD = 64;
void* operator new(size_t n)
{
    void* p = nullptr;
    if (posix_memalign(&p, 64, n + GUARD) != 0 || !p)
        throw std::bad_alloc();
    memset(p, 0, GUARD);
    return (char*)p + GUARD;
}
void* operator new[](size_t n) { return operator new(n); }
void operator delete(void* p) noexcept
{
    if (p)
        free((char*)p - GUARD);
}
void operator delete[](void* p) noexcept { operator delete(p); }
void operator delete(void* p, size_t) noexcept { operator delete(p); }
void operator delete[](void* p, size_t) noexcept { operator delete(p); }

/* globals of the reference that are not in namespace params */
extern int channels;   /* lidar_segmentation.cpp:4 */
extern int rep;        /* star_shaped_search.cpp:8 */
extern float width;    /* star_shaped_search.cpp:9 */

static void set_params(const urf_params& p)
{
    params::fixedFrame = "lidar";
    params::topicName = "points";
    params::x_zero_method = p.x_zero_method != 0;
    params::z_zero_method = p.z_zero_method != 0;
    params::star_shaped_method = p.star_shaped_method != 0;
    params::blind_spots = p.blind_spots != 0;
    params::xDirection = p.xDirection;
    params::interval = p.interval;
    params::curbHeight = p.curbHeight;
    params::curbPoints = p.curbPoints;
    params::beamZone = p.beamZone;
    params::angleFilter1 = p.angleFilter1;
    params::angleFilter2 = p.angleFilter2;
    params::angleFilter3 = p.angleFilter3;
    params::min_X = p.min_X;
    params::max_X = p.max_X;
    params::min_Y = p.min_Y;
    params::max_Y = p.max_Y;
    params::min_Z = p.min_Z;
    params::max_Z = p.max_Z;
    params::kdev_param = p.kdev_param;
    params::kdist_param = p.kdist_param;
    params::starbeam_filter = p.starbeam_filter != 0;
    params::dmin_param = p.dmin_param;
    channels = p.channels;
    width = p.beam_width;
}

struct Job {
    const char* in_path;
    const char* out_path;
    int rc;
};

static void label_from(const char* topic, uint8_t bits, uint8_t* labels, uint32_t n, uint32_t* count,
                       std::vector<uint32_t>* order = nullptr)
{
    auto it = pcl::shim_store().find(topic);
    if (it == pcl::shim_store().end())
        return;
    for (const pcl::PointXYZI& p : it->second) {
        uint32_t id = (uint32_t)p.intensity;
        if (id < n)
            labels[id] |= bits;
        if (order)
            order->push_back(id);
    }
    if (count)
        *count = (uint32_t)it->second.size();
}

static void* run(void* arg)
{
    Job* job = (Job*)arg;
    job->rc = 1;
    FILE* f = fopen(job->in_path, "rb");
    if (!f) { perror("open in"); return nullptr; }
    char magic[8];
    uint32_t hdr[4];
    urf_params prm;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "URFREFIN", 8) != 0 ||
        fread(hdr, 4, 4, f) != 4 || fread(&prm, sizeof(prm), 1, f) != 1 || prm.size != sizeof(prm)) {
        fprintf(stderr, "bad input header\n");
        return nullptr;
    }
    const uint32_t n_scans = hdr[0], n = hdr[1], repeat = hdr[2] ? hdr[2] : 1, flags = hdr[3];
    if (flags & 1u) {
        urf_marker_params mp;
        if (fread(&mp, sizeof(mp), 1, f) != 1 || mp.size != sizeof(mp)) { fprintf(stderr, "bad marker params\n"); return nullptr; }
        params::polysimp_allow = mp.simple_poly_allow != 0;   /* main.cpp:29-32 */
        params::polysimp = mp.poly_s_param;
        params::polyz = mp.poly_z_manual;
        params::zavg_allow = mp.poly_z_avg_allow != 0;
    }
    if (prm.sectors != rep) {
        fprintf(stderr, "the reference is compiled for rep=%d sectors\n", rep);
        return nullptr;
    }
    std::vector<std::vector<float>> xyz(n_scans, std::vector<float>(3 * (size_t)n));
    for (uint32_t s = 0; s < n_scans; s++)
        if (fread(xyz[s].data(), 4, 3 * (size_t)n, f) != 3 * (size_t)n) { fprintf(stderr, "short input\n"); return nullptr; }
    fclose(f);

    set_params(prm);
    ros::NodeHandle nh;
    Detector det(&nh);   /* lidar_segmentation.cpp:51-65, runs beam_init() */

    std::vector<std::vector<uint8_t>> labels(n_scans, std::vector<uint8_t>(n, 0));
    std::vector<urf_scan_info> infos(n_scans);
    std::vector<std::vector<uint32_t>> o_road(n_scans), o_curb(n_scans), o_r10(n_scans);
    std::vector<visualization_msgs::MarkerArray> o_ma(n_scans);
    std::vector<uint32_t> o_pub(n_scans, 0);
    pcl::PointCloud<pcl::PointXYZI> cloud;
    cloud.points.resize(n);
    double ms_first = 0, ms_sum = 0;
    long timed = 0;
    bool first = true;
    for (uint32_t r = 0; r < repeat; r++) {
        for (uint32_t s = 0; s < n_scans; s++) {
            const float* X = xyz[s].data();
            const float* Y = X + n;
            const float* Z = Y + n;
            for (uint32_t i = 0; i < n; i++) {
                cloud.points[i].x = X[i];
                cloud.points[i].y = Y[i];
                cloud.points[i].z = Z[i];
                cloud.points[i].intensity = (float)i;
            }
            pcl::shim_store().clear();
            delete visualization_msgs::shim_markers();
            visualization_msgs::shim_markers() = nullptr;
            auto t0 = std::chrono::steady_clock::now();
            det.filtered(cloud);   /* the reference callback, lidar_segmentation.cpp:95 */
            auto t1 = std::chrono::steady_clock::now();
            double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
            if (first) { ms_first = ms; first = false; }
            else { ms_sum += ms; timed++; }
            if (r == 0) {
                urf_scan_info& in = infos[s];
                memset(&in, 0, sizeof(in));
                uint8_t* L = labels[s].data();
                if (pcl::shim_store().empty()) {
                    in.status = URF_TOO_FEW_POINTS;   /* nothing published, lidar_segmentation.cpp:124-126 */
                } else {
                    label_from("roi", URF_FLAG_ROI, L, n, &in.n_roi);
                    label_from("road", URF_LABEL_ROAD, L, n, &in.n_road, &o_road[s]);
                    label_from("curb", URF_LABEL_CURB, L, n, &in.n_curb, &o_curb[s]);
                    label_from("road_probably", URF_FLAG_RING10, L, n, &in.n_ring10, &o_r10[s]);
                }
                if (visualization_msgs::shim_markers()) {
                    o_pub[s] = 1;
                    o_ma[s] = *visualization_msgs::shim_markers();
                }
            }
        }
    }
    double ms_steady = timed ? ms_sum / (double)timed : ms_first;

    FILE* o = fopen(job->out_path, "wb");
    if (!o) { perror("open out"); return nullptr; }
    fwrite("URFREFOU", 1, 8, o);
    uint32_t oh[2] = { n_scans, n };
    fwrite(oh, 4, 2, o);
    fwrite(&ms_steady, 8, 1, o);
    fwrite(&ms_first, 8, 1, o);
    for (uint32_t s = 0; s < n_scans; s++) {
        fwrite(&infos[s], sizeof(urf_scan_info), 1, o);
        fwrite(labels[s].data(), 1, n, o);
        fwrite(o_road[s].data(), 4, o_road[s].size(), o);
        fwrite(o_curb[s].data(), 4, o_curb[s].size(), o);
        fwrite(o_r10[s].data(), 4, o_r10[s].size(), o);
        if (flags & 1u) {
            const uint32_t nm = (uint32_t)o_ma[s].markers.size();
            fwrite(&o_pub[s], 4, 1, o);
            fwrite(&nm, 4, 1, o);
            for (const visualization_msgs::Marker& m : o_ma[s].markers) {
                const int32_t ia[3] = { m.id, m.action, m.type };
                const float col[4] = { m.color.r, m.color.g, m.color.b, m.color.a };
                const uint32_t np = (uint32_t)m.points.size();
                fwrite(ia, 4, 3, o);
                fwrite(col, 4, 4, o);
                fwrite(&np, 4, 1, o);
                for (const geometry_msgs::Point& q : m.points) {
                    const double xyz[3] = { q.x, q.y, q.z };
                    fwrite(xyz, 8, 3, o);
                }
            }
        }
    }
    fclose(o);
    job->rc = 0;
    return nullptr;
}

int main(int argc, char** argv)
{
    if (argc != 3) {
        fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
        return 2;
    }
    Job job = { argv[1], argv[2], 1 };
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, (size_t)1 << 30);   /* lidar_segmentation.cpp:298 VLA */
    pthread_t th;
    if (pthread_create(&th, &at, run, &job) != 0) { perror("pthread_create"); return 1; }
    pthread_join(th, nullptr);
    return job.rc;
}
