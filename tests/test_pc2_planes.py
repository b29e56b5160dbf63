"""The host-side gather of the callback path (urf_pc2_to_planes: what urf_classify_pc2_async does with a message
it stages) against numpy, without a GPU: every record layout class -- x y z side by side with a fourth word behind
them (four records at a time through an SSE transpose: streaming stores into 16-byte aligned planes, ordinary ones
otherwise), side by side at the record's end, scattered, overlapping --, lengths that are no multiple of four,
NaN / inf / denormal payloads (bit patterns must survive)."""
import numpy as np
import pytest

import urban_road_filter_amd as u

LAYOUTS = [(32, 0, 4, 8), (16, 0, 4, 8), (12, 0, 4, 8), (16, 4, 8, 12), (20, 4, 8, 12), (48, 20, 4, 36), (16, 8, 4, 0), (4, 0, 0, 0),
           (22, 1, 5, 9), (64, 48, 52, 56)]


@pytest.mark.parametrize("step,ox,oy,oz", LAYOUTS)
def test_gather_equals_numpy(step, ox, oy, oz):
    rng = np.random.default_rng(step * 1000 + ox)
    for n in (0, 1, 3, 4, 5, 63, 64, 1001, 4096, 32771):
        raw = rng.integers(0, 256, size=n * step + 3, dtype=np.uint8)   # every bit pattern, NaNs and denormals included
        for shift in (0, 1):                                             # a message that starts at an odd address
            msg = raw[shift:shift + n * step]
            want = [np.ascontiguousarray(msg.reshape(n, step)[:, o:o + 4]).view(np.uint32).reshape(-1) if n else np.zeros(0, np.uint32)
                    for o in (ox, oy, oz)]
            x, y, z = u.pc2_to_planes(msg, n, step, ox, oy, oz)
            for got, w in zip((x, y, z), want):
                assert np.array_equal(got.view(np.uint32), w), (n, shift)
            # destinations that are not 16-byte aligned (the pinned planes of the library are)
            pool = np.empty(3 * (n + 8) + 1, np.float32)
            outs = tuple(pool[1 + k * (n + 8):1 + k * (n + 8) + n] for k in range(3))
            u.pc2_to_planes(msg, n, step, ox, oy, oz, out=outs)
            for got, w in zip(outs, want):
                assert np.array_equal(got.view(np.uint32), w), (n, shift, "unaligned")


def test_layouts_outside_the_record_are_refused():
    msg = np.zeros(64, np.uint8)
    for step, ox, oy, oz in ((16, 13, 0, 4), (16, 0, 0xFFFFFFFE, 4), (3, 0, 0, 0), (16, 0, 4, 14)):
        with pytest.raises(u.UrfError) as e:
            u.pc2_to_planes(msg, 2, step, ox, oy, oz)
        assert e.value.code == -1
