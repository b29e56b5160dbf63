"""Known-answer tests for the polygon simplification.

The reference calls boost::geometry::simplify(line, out, poly_s_param) at
src/lidar_segmentation.cpp:475,512,548.  Boost.Geometry is third party, absent from the reference
checkout and from this image, so the step cannot be compared with the real library.  It is pinned
instead by
  * the worked example of Boost.Geometry's own documentation of `simplify`
    (libs/geometry/doc, reference/algorithms/simplify.html, example simplify.cpp): the linestring
    (1.1 1.1, 2.5 2.1, 3.1 3.1, 4.9 1.1, 3.1 1.9) simplified with distance 0.5 gives
    (1.1 1.1, 3.1 3.1, 4.9 1.1, 3.1 1.9);
  * hand-derived cases for the documented strategy (Douglas-Peucker with the projected-point =
    point-to-SEGMENT distance; a point is kept iff its distance is strictly greater than
    max_distance; fewer than 3 points: unchanged),
applied to BOTH restatements: the oracle's (oracle/urf_rdp.c, used by oracle A's stand-in header
and by oracle B) and the product's (urf::simplifyLine in csrc/marker.cpp through urf_simplify_line)."""
import ctypes as C

import numpy as np
import pytest

import oracles as O
import urban_road_filter_amd as u

KATS = [
    # name, points, max_distance, indices kept
    ("boost_doc_example", [(1.1, 1.1), (2.5, 2.1), (3.1, 3.1), (4.9, 1.1), (3.1, 1.9)], 0.5, [0, 2, 3, 4]),
    ("collinear", [(0, 0), (1, 0), (2, 0), (3, 0), (7, 0)], 0.1, [0, 4]),
    ("exactly_at_the_tolerance_is_dropped", [(0, 0), (1, 0.5), (2, 0)], 0.5, [0, 2]),            # strictly greater
    ("just_above_the_tolerance_is_kept", [(0, 0), (1, 0.5), (2, 0)], 0.4999, [0, 1, 2]),
    ("two_points", [(0, 0), (5, 5)], 0.5, [0, 1]),
    ("one_point", [(3, 4)], 0.5, [0]),
    ("negative_distance_copies", [(0, 0), (1, 0), (2, 0)], -1.0, [0, 1, 2]),
    # distance to the SEGMENT, not to the infinite line: (6, 0.3) is 0.3 from the line through
    # (0,0)-(4,0) but 2.02 from the segment
    ("segment_not_line", [(0, 0), (6, 0.3), (4, 0)], 0.5, [0, 1, 2]),
    # recursion: the farthest point (2,2) splits the span, each half is judged against its own chord:
    # (1,0.2) is 0.8 / sqrt(2) = 0.566 from the chord (0,0)-(2,2)
    ("recursion_drops_the_halves", [(0, 0), (1, 0.2), (2, 2), (3, 0.2), (4, 0)], 0.6, [0, 2, 4]),
    ("recursion_keeps_the_halves", [(0, 0), (1, 0.2), (2, 2), (3, 0.2), (4, 0)], 0.5, [0, 1, 2, 3, 4]),
    ("zero_distance_keeps_every_bend", [(0, 0), (1, 1), (2, 0), (3, 1)], 0.0, [0, 1, 2, 3]),
    ("closed_ring", [(0, 0), (2, 0), (2, 2), (0, 2), (0, 0)], 1.5, [0, 2, 4]),
]


def oracle_keep(pts, d):
    L = C.CDLL(O.ORACLE_B)
    x = np.array([p[0] for p in pts], np.float32)
    y = np.array([p[1] for p in pts], np.float32)
    keep = np.zeros(len(pts), np.uint8)
    L.urf_rdp_float.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    L.urf_rdp_float.restype = None
    L.urf_rdp_float(x.ctypes.data, y.ctypes.data, len(pts), d, keep.ctypes.data)
    return np.flatnonzero(keep).tolist()


def product_keep(pts, d):
    L = u.lib()
    xy = np.array(pts, np.float32).reshape(-1)
    keep = np.zeros(len(pts), np.uint8)
    L.urf_simplify_line.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p]
    assert L.urf_simplify_line(xy.ctypes.data, len(pts), d, keep.ctypes.data) == 0
    return np.flatnonzero(keep).tolist()


@pytest.mark.parametrize("name,pts,d,want", KATS, ids=[k[0] for k in KATS])
def test_known_answers(name, pts, d, want):
    O.ensure_built()
    assert oracle_keep(pts, d) == want
    assert product_keep(pts, d) == want


def test_closed_ring_farthest_of_a_degenerate_chord():
    """first == last: the chord is a point, every distance is the distance to that point; the
    farthest corner (2,2) is kept, then each half is judged against its own chord: (2,0) and (0,2)
    are sqrt(2) from their chords -> kept as well with a smaller tolerance."""
    ring = [(0, 0), (2, 0), (2, 2), (0, 2), (0, 0)]
    assert oracle_keep(ring, 1.0) == product_keep(ring, 1.0) == [0, 1, 2, 3, 4]


def test_both_restatements_agree_on_random_lines():
    rng = np.random.default_rng(5)
    for k in range(200):
        n = int(rng.integers(1, 60))
        pts = [tuple(map(float, p)) for p in (rng.standard_normal((n, 2)) * 3).astype(np.float32)]
        d = float(np.float32(rng.random() * 2))
        assert oracle_keep(pts, d) == product_keep(pts, d)
