/*
 * params.cpp -- parameter defaults, validation and status strings of the C ABI.
 * Host code.
 */
#include <cstddef>
#include <cstring>

#include "urf.h"
#include "urf_internal.hpp"

/* cfg/LidarFilters.cfg:10-84 defaults as stored by paramsCallback (src/main.cpp:4-34)
 * into the float/int/bool globals of namespace params (data_structures.hpp:66-88),
 * plus channels (lidar_segmentation.cpp:4), rep and width (star_shaped_search.cpp:8-9). */
extern "C" int urf_default_params(urf_params* p)
{
    if (!p)
        return URF_ERR_INVALID_ARG;
    std::memset(p, 0, sizeof(*p));
    p->size = sizeof(urf_params);
    p->x_zero_method = 1;        /* cfg:16 */
    p->z_zero_method = 1;        /* cfg:17 */
    p->star_shaped_method = 1;   /* cfg:18 */
    p->blind_spots = 1;          /* cfg:19 */
    p->xDirection = 0;           /* cfg:27 */
    p->interval = 0.1800f;       /* cfg:30 */
    p->curbHeight = 0.0500f;     /* cfg:33 */
    p->curbPoints = 5;           /* cfg:36 */
    p->beamZone = 30.0f;         /* cfg:39 */
    p->min_X = 0.0f;             /* cfg:42 */
    p->max_X = 30.0f;            /* cfg:43 */
    p->min_Y = -10.0f;           /* cfg:46 */
    p->max_Y = 10.0f;            /* cfg:47 */
    p->min_Z = -3.0f;            /* cfg:50 */
    p->max_Z = -1.0f;            /* cfg:51 */
    p->angleFilter1 = 150.0f;    /* cfg:54 cylinder_deg_x */
    p->angleFilter2 = 140.0f;    /* cfg:57 cylinder_deg_z */
    p->angleFilter3 = 50.0f;     /* cfg:60 curb_slope_deg */
    p->kdev_param = 1.225f;      /* cfg:63 */
    p->kdist_param = 2.0f;       /* cfg:66 */
    p->starbeam_filter = 0;      /* cfg:69 */
    p->dmin_param = 10;          /* cfg:72 */
    p->channels = 64;            /* lidar_segmentation.cpp:4 */
    p->sectors = 360;            /* star_shaped_search.cpp:8 */
    p->beam_width = 0.2f;        /* star_shaped_search.cpp:9 */
    return URF_OK;
}

extern "C" int urf_default_marker_params(urf_marker_params* p)
{
    if (!p)
        return URF_ERR_INVALID_ARG;
    p->size = sizeof(urf_marker_params);
    p->simple_poly_allow = 1;    /* cfg:75 */
    p->poly_s_param = 0.7f;      /* cfg:78 */
    p->poly_z_manual = -1.5f;    /* cfg:81 */
    p->poly_z_avg_allow = 1;     /* cfg:84 */
    return URF_OK;
}

int urf_validate_params(const urf_params* p)
{
    if (!p || p->size != sizeof(urf_params))
        return URF_ERR_INVALID_ARG;
    if (p->channels < 1 || p->channels > URF_MAX_CHANNELS)
        return URF_ERR_PARAMS;
    if (p->curbPoints < 1 || p->curbPoints > URF_MAX_CURB_POINTS)   /* cfg:36 range 1..30 */
        return URF_ERR_PARAMS;
    if (p->sectors < 1 || p->sectors > URF_MAX_SECTORS)
        return URF_ERR_PARAMS;
    if (p->xDirection < 0 || p->xDirection > 2)                      /* cfg:27 enum */
        return URF_ERR_PARAMS;
    if (!(p->beamZone > 0.0f) || !(p->beamZone <= 360.0f))           /* cfg:39 range 10..100 */
        return URF_ERR_PARAMS;
    if (!(p->interval >= 0.0f))
        return URF_ERR_PARAMS;
    return URF_OK;
}

/* ---- the live parameter surface -------------------------------------------------
 * cfg/LidarFilters.cfg:10-84, one row per gen.add(): name, type, default, range (dynamic_reconfigure
 * fills in 0..1 for bool_t), the enum of xDirection (:22-27), and where the value lives here. */
#define P_(f) "" #f, URF_PARAM_IN_PARAMS, (uint32_t)offsetof(urf_params, f)
#define M_(f) "" #f, URF_PARAM_IN_MARKER_PARAMS, (uint32_t)offsetof(urf_marker_params, f)
static const urf_param_desc g_param_table[] = {
    /* cfg name            member                 type               default  min     max   string default / enum */
    { "fixed_frame", "", URF_PARAM_NODE_ONLY, 0, URF_PARAM_STR, 0, 0, 0, "left_os1/os1_lidar", nullptr, 10 },
    { "topic_name", "", URF_PARAM_NODE_ONLY, 0, URF_PARAM_STR, 0, 0, 0, "/left_os1/os1_cloud_node/points", nullptr, 13 },
    { "x_zero_method", P_(x_zero_method), URF_PARAM_BOOL, 1, 0, 1, nullptr, nullptr, 16 },
    { "z_zero_method", P_(z_zero_method), URF_PARAM_BOOL, 1, 0, 1, nullptr, nullptr, 17 },
    { "star_shaped_method", P_(star_shaped_method), URF_PARAM_BOOL, 1, 0, 1, nullptr, nullptr, 18 },
    { "blind_spots", P_(blind_spots), URF_PARAM_BOOL, 1, 0, 1, nullptr, nullptr, 19 },
    { "xDirection", P_(xDirection), URF_PARAM_INT, 0, 0, 2, nullptr, "bothX=0,positiveX=1,negativeX=2", 27 },
    { "interval", P_(interval), URF_PARAM_DOUBLE, 0.1800, 0.0100, 10, nullptr, nullptr, 30 },
    { "curb_height", P_(curbHeight), URF_PARAM_DOUBLE, 0.0500, 0.0100, 0.5000, nullptr, nullptr, 33 },
    { "curb_points", P_(curbPoints), URF_PARAM_INT, 5, 1, 30, nullptr, nullptr, 36 },
    { "beamZone", P_(beamZone), URF_PARAM_DOUBLE, 30, 10, 100, nullptr, nullptr, 39 },
    { "min_x", P_(min_X), URF_PARAM_DOUBLE, 0, -200, 200, nullptr, nullptr, 42 },
    { "max_x", P_(max_X), URF_PARAM_DOUBLE, 30, -200, 200, nullptr, nullptr, 43 },
    { "min_y", P_(min_Y), URF_PARAM_DOUBLE, -10, -200, 200, nullptr, nullptr, 46 },
    { "max_y", P_(max_Y), URF_PARAM_DOUBLE, 10, -200, 200, nullptr, nullptr, 47 },
    { "min_z", P_(min_Z), URF_PARAM_DOUBLE, -3, -200, 200, nullptr, nullptr, 50 },
    { "max_z", P_(max_Z), URF_PARAM_DOUBLE, -1, -200, 200, nullptr, nullptr, 51 },
    { "cylinder_deg_x", P_(angleFilter1), URF_PARAM_DOUBLE, 150, 0, 180, nullptr, nullptr, 54 },
    { "cylinder_deg_z", P_(angleFilter2), URF_PARAM_DOUBLE, 140, 0, 180, nullptr, nullptr, 57 },
    { "curb_slope_deg", P_(angleFilter3), URF_PARAM_DOUBLE, 50, 0, 180, nullptr, nullptr, 60 },
    { "kdev_param", P_(kdev_param), URF_PARAM_DOUBLE, 1.225, 0.5, 5, nullptr, nullptr, 63 },
    { "kdist_param", P_(kdist_param), URF_PARAM_DOUBLE, 2, 0.4, 10, nullptr, nullptr, 66 },
    { "starbeam_filter", P_(starbeam_filter), URF_PARAM_BOOL, 0, 0, 1, nullptr, nullptr, 69 },
    { "dmin_param", P_(dmin_param), URF_PARAM_INT, 10, 3, 30, nullptr, nullptr, 72 },
    { "simple_poly_allow", M_(simple_poly_allow), URF_PARAM_BOOL, 1, 0, 1, nullptr, nullptr, 75 },
    { "poly_s_param", M_(poly_s_param), URF_PARAM_DOUBLE, 0.7, 0, 1, nullptr, nullptr, 78 },
    { "poly_z_manual", M_(poly_z_manual), URF_PARAM_DOUBLE, -1.5, -5, 5, nullptr, nullptr, 81 },
    { "poly_z_avg_allow", M_(poly_z_avg_allow), URF_PARAM_BOOL, 1, 0, 1, nullptr, nullptr, 84 },
};
#undef P_
#undef M_

extern "C" int urf_param_count(void) { return (int)(sizeof(g_param_table) / sizeof(g_param_table[0])); }
extern "C" const urf_param_desc* urf_param_table(void) { return g_param_table; }

/* dynamic_reconfigure's server clamps a request to [min, max] before the callback sees it
 * (src/main.cpp:4-34 then stores double -> float, int -> int, bool -> bool). */
extern "C" int urf_clamp_params(urf_params* p, urf_marker_params* mp, uint32_t* n_clamped)
{
    if (!p || p->size != sizeof(urf_params) || (mp && mp->size != sizeof(urf_marker_params)))
        return URF_ERR_INVALID_ARG;
    uint32_t changed = 0;
    for (const urf_param_desc& d : g_param_table) {
        unsigned char* base = d.where == URF_PARAM_IN_PARAMS ? (unsigned char*)p : d.where == URF_PARAM_IN_MARKER_PARAMS ? (unsigned char*)mp : nullptr;
        if (!base)
            continue;
        if (d.type == URF_PARAM_DOUBLE) {
            float v;
            std::memcpy(&v, base + d.offset, sizeof(v));
            float w = v;
            if (!(w >= (float)d.min))   /* NaN goes to the lower bound */
                w = (float)d.min;
            if (w > (float)d.max)
                w = (float)d.max;
            if (std::memcmp(&w, &v, sizeof(v)) != 0) {
                std::memcpy(base + d.offset, &w, sizeof(w));
                changed++;
            }
        } else {
            int32_t v;
            std::memcpy(&v, base + d.offset, sizeof(v));
            int32_t w = v;
            if (d.type == URF_PARAM_BOOL)
                w = v != 0;
            else
                w = v < (int32_t)d.min ? (int32_t)d.min : (v > (int32_t)d.max ? (int32_t)d.max : v);
            if (w != v) {
                std::memcpy(base + d.offset, &w, sizeof(w));
                changed++;
            }
        }
    }
    if (n_clamped)
        *n_clamped = changed;
    return URF_OK;
}

extern "C" const char* urf_strerror(int status)
{
    switch (status) {
    case URF_OK: return "ok";
    case URF_TOO_FEW_POINTS: return "fewer than 30 points in the region of interest: nothing to publish";
    case URF_ERR_INVALID_ARG: return "invalid argument";
    case URF_ERR_NO_DEVICE: return "no usable HIP device";
    case URF_ERR_HIP: return "HIP runtime error";
    case URF_ERR_CAPACITY: return "scan or batch exceeds the capacity given to urf_create";
    case URF_ERR_OOM: return "out of device memory";
    case URF_ERR_PARAMS: return "parameter outside the supported range";
    case URF_ERR_BUSY: return "both slots of the asynchronous single-scan path are in flight";
    default: return "unknown status";
    }
}

extern "C" int urf_abi_version(void) { return URF_ABI_VERSION; }
