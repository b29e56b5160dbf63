/*
 * urf_stdsort.h -- libstdc++'s std::sort, restated in C for the star-shaped search's
 * `polar` records (star_shaped_search.cpp:109: std::sort(p.begin(), p.end(), ptcmpr),
 * ptcmpr(a, b) = a.r < b.r, :22-25).
 *
 * TEST INFRASTRUCTURE ONLY (part of oracle B, see urf_oracle.h).
 *
 * Why: std::sort is not stable, the order it leaves equal planar ranges in is
 * "unspecified" by the standard -- but it is a deterministic function of the input
 * sequence for a given library, the walk of star_shaped_search.cpp:123-149 divides by
 * the difference of neighbouring ranges (a tie gives +-inf or NaN, and which of two
 * tied points comes second decides the sign), so the order IS label-relevant and the
 * reference binary gives ONE answer.  This file follows that answer.
 *
 * The third-party algorithm: libstdc++ (GCC 11.4 in this image; the file:line below
 * are /usr/include/c++/11/bits/stl_algo.h and stl_heap.h; the algorithm is unchanged
 * since GCC 4.5 introduced __move_median_to_first -- the compilers of ROS Kinetic /
 * Melodic / Noetic (GCC 5 / 7 / 9) carry the same code):
 *   __sort                      stl_algo.h:1947-1959   introsort loop, then final insertion sort
 *   __introsort_loop            stl_algo.h:1923-1943   depth limit 2 * floor(log2 n); segments <= 16 are left alone
 *   __unguarded_partition_pivot stl_algo.h:1896-1906   median of (first+1, mid, last-1) moved to first
 *   __move_median_to_first      stl_algo.h:76-97
 *   __unguarded_partition       stl_algo.h:1874-1892   Hoare partition against *first
 *   __partial_sort / heap       stl_algo.h:1909-1918, stl_heap.h:128-146, 219-249, 251-266, 337-360, 416-427
 *   __final_insertion_sort      stl_algo.h:1859-1871   (__insertion_sort :1815-1836, __unguarded_linear_insert :1795-1812)
 * Pinned by tests/test_stdsort.py: this restatement against std::sort itself (oracle A's
 * build of the reference links the real one; oracle/ref_harness.cpp also exposes a
 * "sort only" mode) on random, tie-heavy and median-of-three-killer sequences.
 */
#ifndef URF_STDSORT_H
#define URF_STDSORT_H

typedef struct {
    int id;
    float r;
    float fi;
} urf_polar;   /* data_structures.hpp:51-56 */

#define URF_SS_LESS(a, b) ((a).r < (b).r)   /* star_shaped_search.cpp:22-25 */
#define URF_SS_THRESHOLD 16                 /* stl_algo.h:1855 */

static long urf_ss_heap_sorts;   /* how often the depth limit was reached (tests only: is the fallback exercised?) */

static void urf_ss_swap(urf_polar* a, urf_polar* b)
{
    urf_polar t = *a;
    *a = *b;
    *b = t;
}

/* stl_heap.h:128-146 (__comp(parent, value)) */
static void urf_ss_push_heap(urf_polar* first, long hole, long top, urf_polar value)
{
    long parent = (hole - 1) / 2;
    while (hole > top && URF_SS_LESS(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

/* stl_heap.h:219-249 */
static void urf_ss_adjust_heap(urf_polar* first, long hole, long len, urf_polar value)
{
    const long top = hole;
    long second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (URF_SS_LESS(first[second], first[second - 1]))
            second--;
        first[hole] = first[second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        first[hole] = first[second - 1];
        hole = second - 1;
    }
    urf_ss_push_heap(first, hole, top, value);
}

/* stl_algo.h:1909-1918 with middle == last: __heap_select = __make_heap (stl_heap.h:337-360), then __sort_heap (:416-427) */
static void urf_ss_heap_sort(urf_polar* first, urf_polar* last)
{
    const long len = last - first;
    if (len >= 2) {
        long parent = (len - 2) / 2;
        for (;;) {
            urf_polar value = first[parent];
            urf_ss_adjust_heap(first, parent, len, value);
            if (parent == 0)
                break;
            parent--;
        }
    }
    while (last - first > 1) {
        --last;
        /* __pop_heap(first, last, last): stl_heap.h:251-266 */
        urf_polar value = *last;
        *last = *first;
        urf_ss_adjust_heap(first, 0, last - first, value);
    }
}

/* stl_algo.h:76-97 */
static void urf_ss_move_median_to_first(urf_polar* result, urf_polar* a, urf_polar* b, urf_polar* c)
{
    if (URF_SS_LESS(*a, *b)) {
        if (URF_SS_LESS(*b, *c))
            urf_ss_swap(result, b);
        else if (URF_SS_LESS(*a, *c))
            urf_ss_swap(result, c);
        else
            urf_ss_swap(result, a);
    } else if (URF_SS_LESS(*a, *c))
        urf_ss_swap(result, a);
    else if (URF_SS_LESS(*b, *c))
        urf_ss_swap(result, c);
    else
        urf_ss_swap(result, b);
}

/* stl_algo.h:1874-1892 */
static urf_polar* urf_ss_unguarded_partition(urf_polar* first, urf_polar* last, urf_polar* pivot)
{
    for (;;) {
        while (URF_SS_LESS(*first, *pivot))
            ++first;
        --last;
        while (URF_SS_LESS(*pivot, *last))
            --last;
        if (!(first < last))
            return first;
        urf_ss_swap(first, last);
        ++first;
    }
}

/* stl_algo.h:1923-1943 */
static void urf_ss_introsort_loop(urf_polar* first, urf_polar* last, long depth_limit)
{
    while (last - first > URF_SS_THRESHOLD) {
        if (depth_limit == 0) {
            urf_ss_heap_sorts++;
            urf_ss_heap_sort(first, last);
            return;
        }
        --depth_limit;
        urf_polar* mid = first + (last - first) / 2;   /* :1896-1906 */
        urf_ss_move_median_to_first(first, first + 1, mid, last - 1);
        urf_polar* cut = urf_ss_unguarded_partition(first + 1, last, first);
        urf_ss_introsort_loop(cut, last, depth_limit);
        last = cut;
    }
}

/* stl_algo.h:1795-1812 (__comp(val, next)) */
static void urf_ss_unguarded_linear_insert(urf_polar* last)
{
    urf_polar val = *last;
    urf_polar* next = last;
    --next;
    while (URF_SS_LESS(val, *next)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}

/* stl_algo.h:1815-1836 */
static void urf_ss_insertion_sort(urf_polar* first, urf_polar* last)
{
    if (first == last)
        return;
    for (urf_polar* i = first + 1; i != last; ++i) {
        if (URF_SS_LESS(*i, *first)) {
            urf_polar val = *i;
            for (urf_polar* q = i; q != first; --q)   /* move_backward(first, i, i + 1) */
                *q = *(q - 1);
            *first = val;
        } else
            urf_ss_unguarded_linear_insert(i);
    }
}

/* stl_algo.h:1947-1959 and :1859-1871 */
static void urf_std_sort_polar(urf_polar* first, long n)
{
    urf_polar* last = first + n;
    if (n == 0)
        return;
    long lg = 0;   /* std::__lg(n) = floor(log2(n)) */
    while ((n >> (lg + 1)) != 0)
        lg++;
    urf_ss_introsort_loop(first, last, lg * 2);
    if (n > URF_SS_THRESHOLD) {
        urf_ss_insertion_sort(first, first + URF_SS_THRESHOLD);
        for (urf_polar* i = first + URF_SS_THRESHOLD; i != last; ++i)   /* __unguarded_insertion_sort :1839-1848 */
            urf_ss_unguarded_linear_insert(i);
    } else
        urf_ss_insertion_sort(first, last);
}

#endif /* URF_STDSORT_H */
