/*
 * params.cpp -- parameter defaults, validation and status strings of the C ABI.
 * Host code.
 */
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
    p->simple_poly_allow = 1;    /* cfg:76 */
    p->poly_s_param = 0.7f;      /* cfg:79 */
    p->poly_z_manual = -1.5f;    /* cfg:82 */
    p->poly_z_avg_allow = 1;     /* cfg:85 */
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
    default: return "unknown status";
    }
}

extern "C" int urf_abi_version(void) { return URF_ABI_VERSION; }
