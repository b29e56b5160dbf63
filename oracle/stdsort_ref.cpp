/*
 * stdsort_ref.cpp -- the REAL std::sort of this image's libstdc++ behind a C entry point.
 *
 * TEST INFRASTRUCTURE ONLY.  tests/test_stdsort.py holds oracle B's restatement of the
 * algorithm (urf_stdsort.h) against it.  The record and the comparison are the ones of the
 * reference's call (data_structures.hpp:51-56 `polar`, star_shaped_search.cpp:22-25 `ptcmpr`,
 * :109 the call); nothing of the reference is compiled here.
 */
#include <algorithm>
#include <vector>

namespace {
struct polar {
    int id;
    float r;
    float fi;
};
bool by_r(polar a, polar b) { return a.r < b.r; }
}   // namespace

extern "C" void urf_ref_std_sort(float* r, int* id, int n)
{
    std::vector<polar> p((size_t)(n > 0 ? n : 0));
    for (int i = 0; i < n; i++)
        p[i] = polar{ id[i], r[i], 0.f };
    std::sort(p.begin(), p.end(), by_r);
    for (int i = 0; i < n; i++) {
        id[i] = p[i].id;
        r[i] = p[i].r;
    }
}

/* McIlroy, "A Killer Adversary for Quicksort" (1999): std::sort is run on the indices 0..n-1 with a comparison that
 * fixes a value only when it must ("gas" = not yet decided); the values it ends up with are an input on which THIS
 * std::sort degenerates -- i.e. reaches its depth limit and runs the heap sort fallback.  out[i] = value of item i. */
namespace {
int* g_val;
int g_nsolid, g_candidate, g_gas;
bool adversary(int x, int y)
{
    if (g_val[x] == g_gas && g_val[y] == g_gas) {
        if (x == g_candidate)
            g_val[x] = g_nsolid++;
        else
            g_val[y] = g_nsolid++;
    }
    if (g_val[x] == g_gas)
        g_candidate = x;
    else if (g_val[y] == g_gas)
        g_candidate = y;
    return g_val[x] < g_val[y];
}
}   // namespace

extern "C" void urf_ref_killer(float* out, int n)
{
    std::vector<int> val((size_t)n), idx((size_t)n);
    g_val = val.data();
    g_gas = n - 1;
    g_nsolid = g_candidate = 0;
    for (int i = 0; i < n; i++) {
        idx[i] = i;
        val[i] = g_gas;
    }
    std::sort(idx.begin(), idx.end(), adversary);
    for (int i = 0; i < n; i++)
        out[i] = (float)val[i];
}
