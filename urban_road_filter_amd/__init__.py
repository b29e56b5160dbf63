"""urban_road_filter_amd -- MI355X-native road/curb classification of LiDAR sweeps.

The product is the C-ABI shared library ``liburf_hip.so`` (include/urf.h): hand-written
gfx950 HIP kernels behind the reference's ``Detector::filtered`` boundary.  This Python
package is plumbing for tests and the benchmark: a ctypes binding of that ABI.
"""
from .api import (  # noqa: F401
    Context,
    MarkerParams,
    ParamDesc,
    Params,
    ScanInfo,
    UrfError,
    clamp_params,
    default_marker_params,
    default_params,
    param_table,
    PARAM_BOOL, PARAM_INT, PARAM_DOUBLE, PARAM_STR,
    lib,
    lib_path,
    test_lib,
    synth_cloud,
    pc2_to_planes,
    LABEL_MASK, LABEL_ROAD, LABEL_CURB, FLAG_ROI, FLAG_RING, FLAG_RING10,
    STAGE_VALPHA, STAGE_RING, STAGE_AZIMUTH, STAGE_RANGE2D, STAGE_DETECT, STAGE_SECTOR,
    STAGE_ANGLE_TABLE, STAGE_MAXDIST, STAGE_QUADRANTS, STAGE_BEAM_STOP,
)
