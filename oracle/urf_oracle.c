/*
 * urf_oracle.c -- CPU oracle ("oracle B"): a plain-C restatement of the
 * reference's per-scan road/curb classification.
 *
 * TEST INFRASTRUCTURE ONLY (see urf_oracle.h).
 *
 * Every function names the reference lines it follows (paths relative to the
 * reference checkout).  Arithmetic follows SURVEY.md appendix A: which
 * operands are float, which are double, and in which order they are combined;
 * build with -ffp-contract=off (the reference is built -std=c++17 -O2, where
 * GCC contracts nothing).  The libm calls acosf/asinf/atan2f are replaced by
 * include/urf_libm.h (rationale there); sqrt/sqrtf/fabsf are exact IEEE
 * operations and come from the C library.
 *
 * Deliberate deviations from the reference (its behaviour there is undefined
 * or a crash -- SURVEY.md appendix B "define away"):
 *   D1  star sector index == sectors (azimuth in (-5e-7,0) rad) wraps to
 *       sector 0 instead of dereferencing beamp[360] == nullptr
 *       (star_shaped_search.cpp:20,157,171-173);
 *   (D2 of rounds 1-4 -- equal planar ranges of a sector ordered by ROI index -- is gone: the
 *       sector is sorted by a literal restatement of libstdc++'s std::sort, urf_stdsort.h, whose
 *       order of equal ranges is a deterministic function of the input and decides labels;)
 *   D3  array3D[k][-1] / [n] over-reads of blind_spots.cpp:107,216,... are
 *       not performed (the value read is never used);
 *   D4  array3D[1] / array3D[10] are only touched when channels > 1 / > 10
 *       (blind_spots.cpp:19, lidar_segmentation.cpp:605).
 * Points with x == y == 0 (NaN azimuth) are sorted exactly as the reference's
 * Lomuto quicksort sorts them (the HIP path follows: k_nan_rings).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "urf_oracle.h"
#include "urf_libm.h"
#include "urf_rdp.h"
#include "urf_stdsort.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

float urf_oracle_acosf(float x) { return urf_acosf(x); }
float urf_oracle_asinf(float x) { return urf_asinf(x); }
float urf_oracle_atan2f(float y, float x) { return urf_atan2f(y, x); }

/* data_structures.hpp:40-49 (Point2D / Point3D), without the PCL padding */
typedef struct {
    float x, y, z;
    float d;
    float alpha;
    short isCurbPoint;
    float newY;
    uint32_t src;      /* index of the point in the INPUT cloud */
    uint8_t detect;    /* bit0 star, bit1 x_zero, bit2 z_zero (debug only) */
} pt3;

/* data_structures.hpp:51-56 */
typedef urf_polar polar;

/* data_structures.hpp:58-64 */
typedef struct {
    polar* p;
    int n, cap;
    int yx;
    float o, d;
} box;

static void box_push(box* b, polar v)
{
    if (b->n == b->cap) {
        b->cap = b->cap ? 2 * b->cap : 64;
        b->p = (polar*)realloc(b->p, (size_t)b->cap * sizeof(polar));
    }
    b->p[b->n++] = v;
}

/* star_shaped_search.cpp:22 ptcmpr and :109 std::sort: urf_stdsort.h.  Exported for tests/test_stdsort.py, which holds
 * the restatement against the real std::sort (oracle/stdsort_ref.cpp). */
long urf_oracle_std_sort_heap_sorts(void) { return urf_ss_heap_sorts; }
void urf_oracle_std_sort(float* r, int* id, int n)
{
    polar* p = (polar*)malloc((size_t)(n > 0 ? n : 1) * sizeof(polar));
    for (int i = 0; i < n; i++) {
        p[i].id = id[i];
        p[i].r = r[i];
        p[i].fi = 0.f;
    }
    urf_std_sort_polar(p, n);
    for (int i = 0; i < n; i++) {
        id[i] = p[i].id;
        r[i] = p[i].r;
    }
    free(p);
}

/* star_shaped_search.cpp:32-66 beam_init: per-sector constants.  `fi` is a
 * float, so tan/sin/cos(fi) are the float overloads; tan(0.5*M_PI - fi) has a
 * double argument. */
static void beam_init(box* beams, int rep, float width)
{
    float fi, off = (float)(0.5 * (double)width);
    for (int i = 0; i < rep; i++) {
        fi = (float)((double)(i * 2) * M_PI / (double)rep);
        if (fabsf(tanf(fi)) > 1) {
            beams[i].yx = 1;
            beams[i].d = (float)tan(0.5 * M_PI - (double)fi);
            beams[i].o = fabsf(off / sinf(fi));
        } else {
            beams[i].yx = 0;
            beams[i].d = tanf(fi);
            beams[i].o = fabsf(off / cosf(fi));
        }
    }
}

/* star_shaped_search.cpp:68-153 beamfunc */
static void beamfunc(box* bm, pt3* array2D, const urf_params* prm, float slope_param, int16_t* dbg_sector)
{
    int i = 0, s = bm->n;
    float c;

    if (prm->starbeam_filter) {   /* :73-107 */
        int w = 0;
        for (i = 0; i < s; i++) {
            const pt3* q = &array2D[bm->p[i].id];
            int keep;
            if (bm->yx) {
                c = bm->d * q->y;
                keep = ((c - bm->o) < q->x && q->x < (c + bm->o));
            } else {
                c = bm->d * q->x;
                keep = ((c - bm->o) < q->y && q->y < (c + bm->o));
            }
            if (keep)
                bm->p[w++] = bm->p[i];   /* erase() keeps the order of the survivors */
            else if (dbg_sector)
                dbg_sector[q->src] = -1; /* stage output: sector of the points that take part */
        }
        s = bm->n = w;
    }

    urf_std_sort_polar(bm->p, s);   /* :109, libstdc++'s introsort literally (urf_stdsort.h) */

    if (s > 1) {   /* :112-150 */
        float kdev = prm->kdev_param;
        float kdist = prm->kdist_param;
        int dmin = prm->dmin_param;
        float avg = 0, dev = 0, nan = 0;
        float ax, ay, bx, by, slp;
        bx = bm->p[0].r;
        by = array2D[bm->p[0].id].z;
        for (i = 1; i < s; i++) {
            ax = bx;
            bx = bm->p[i].r;
            ay = by;
            by = array2D[bm->p[i].id].z;
            slp = (by - ay) / (bx - ax);   /* :27-30 slope() */
            if (slp != slp) {
                nan++;
            } else {
                avg *= (float)i - nan - 1;
                avg += slp;
                avg *= 1 / ((float)i - nan);
                dev *= (float)i - nan - 1;
                dev += fabsf(slp - avg);
                dev *= 1 / ((float)i - nan);
            }
            if (slp > slope_param ||
                (i > dmin && (slp * slp - avg * avg) * kdev * ((bx - ax) * kdist) > dev)) {
                array2D[bm->p[i].id].isCurbPoint = 2;   /* :146 */
                array2D[bm->p[i].id].detect |= 1;
                break;
            }
        }
    }
    bm->n = 0;   /* :152 */
}

/* star_shaped_search.cpp:155-181 starShapedSearch */
static void star_shaped_search(pt3* array2D, int s, const urf_params* prm, int16_t* dbg_sector)
{
    int rep = prm->sectors;
    box* beams = (box*)calloc((size_t)rep, sizeof(box));
    beam_init(beams, rep, prm->beam_width);
    float Kfi = (float)((double)rep / (2 * M_PI));                          /* :65 */
    float slope_param = (float)((double)prm->angleFilter3 * (M_PI / 180)); /* :160 */
    int f;
    float r, fi;
    for (int i = 0; i < s; i++) {
        r = sqrtf(array2D[i].x * array2D[i].x + array2D[i].y * array2D[i].y); /* :164 */
        fi = urf_atan2f(array2D[i].y, array2D[i].x);                          /* :166 */
        if (fi < 0)
            fi = (float)((double)fi + 2 * M_PI);                               /* :168-169 */
        f = (int)(fi * Kfi);                                                  /* :171 */
        if (f >= rep)
            f = 0;                                                            /* deviation D1 */
        polar pl = { i, r, fi };
        box_push(&beams[f], pl);                                              /* :173 */
        if (dbg_sector)
            dbg_sector[array2D[i].src] = (int16_t)f;
    }
    for (int i = 0; i < rep; i++)
        beamfunc(&beams[i], array2D, prm, slope_param, dbg_sector);                     /* :177-180 */
    for (int i = 0; i < rep; i++)
        free(beams[i].p);
    free(beams);
}

/* x_zero_method.cpp:7-71 */
static void x_zero_method(pt3** array3D, int index, const int* indexArray, const urf_params* prm)
{
    int p2, p3;
    float alpha, x1, x2, x3, d, bracket;
    const int cp = prm->curbPoints;
    for (int i = 0; i < index; i++) {
        pt3* a = array3D[i];
        for (int j = 1; j < indexArray[i]; j++)
            a[j].newY = (float)((double)a[j - 1].newY + 0.0100);   /* :26 */

        for (int j = cp; j <= (indexArray[i] - 1) - cp; j++) {     /* :30 */
            p2 = j + cp / 2;
            p3 = j + cp;
            {
                double dx = (double)(a[p3].x - a[j].x), dy = (double)(a[p3].y - a[j].y);
                d = (float)sqrt(dx * dx + dy * dy);                 /* :35-37 */
            }
            if ((double)d < 5.0000) {                               /* :40 */
                double u, v;
                u = (double)(a[p2].newY - a[j].newY);  v = (double)(a[p2].z - a[j].z);
                x1 = (float)sqrt(u * u + v * v);                    /* :42-44 */
                u = (double)(a[p3].newY - a[p2].newY); v = (double)(a[p3].z - a[p2].z);
                x2 = (float)sqrt(u * u + v * v);                    /* :45-47 */
                u = (double)(a[p3].newY - a[j].newY);  v = (double)(a[p3].z - a[j].z);
                x3 = (float)sqrt(u * u + v * v);                    /* :48-50 */

                {
                    double num = (double)x3 * (double)x3 - (double)x1 * (double)x1 - (double)x2 * (double)x2;
                    float den = (-2 * x1) * x2;                     /* int*float*float, left to right */
                    bracket = (float)(num / (double)den);           /* :52 */
                }
                if (bracket < -1)
                    bracket = -1;
                else if (bracket > 1)
                    bracket = 1;

                alpha = (float)((double)(urf_acosf(bracket) * 180) / M_PI);   /* :58 */

                if (alpha <= prm->angleFilter1 &&
                    (fabsf(a[j].z - a[p2].z) >= prm->curbHeight ||
                     fabsf(a[p3].z - a[p2].z) >= prm->curbHeight) &&
                    (double)fabsf(a[j].z - a[p3].z) >= 0.05) {     /* :61-64 */
                    a[p2].isCurbPoint = 2;                          /* :66 */
                    a[p2].detect |= 2;
                }
            }
        }
    }
}

/* z_zero_method.cpp:5-76 */
static void z_zero_method(pt3** array3D, int index, const int* indexArray, const urf_params* prm)
{
    float alpha, va1, va2, vb1, vb2, max1, max2, d, bracket;
    const int cp = prm->curbPoints;
    for (int i = 0; i < index; i++) {
        pt3* a = array3D[i];
        for (int j = cp; j <= (indexArray[i] - 1) - cp; j++) {     /* :21 */
            {
                double dx = (double)(a[j + cp].x - a[j - cp].x), dy = (double)(a[j + cp].y - a[j - cp].y);
                d = (float)sqrt(dx * dx + dy * dy);                 /* :23-25 */
            }
            if ((double)d < 5.0000) {                               /* :28 */
                max1 = max2 = fabsf(a[j].z);
                va1 = va2 = vb1 = vb2 = 0;
                for (int k = j - 1; k >= j - cp; k--) {             /* :35-41 */
                    va1 = va1 + (a[k].x - a[j].x);
                    va2 = va2 + (a[k].y - a[j].y);
                    if (fabsf(a[k].z) > max1)
                        max1 = fabsf(a[k].z);
                }
                for (int k = j + 1; k <= j + cp; k++) {             /* :44-50 */
                    vb1 = vb1 + (a[k].x - a[j].x);
                    vb2 = vb2 + (a[k].y - a[j].y);
                    if (fabsf(a[k].z) > max2)
                        max2 = fabsf(a[k].z);
                }
                va1 = (1 / (float)cp) * va1;                        /* :52-55 */
                va2 = (1 / (float)cp) * va2;
                vb1 = (1 / (float)cp) * vb1;
                vb2 = (1 / (float)cp) * vb2;

                {
                    float num = va1 * vb1 + va2 * vb2;
                    double na = sqrt((double)va1 * (double)va1 + (double)va2 * (double)va2);
                    double nb = sqrt((double)vb1 * (double)vb1 + (double)vb2 * (double)vb2);
                    bracket = (float)((double)num / (na * nb));     /* :57 */
                }
                if (bracket < -1)
                    bracket = -1;
                else if (bracket > 1)
                    bracket = 1;

                alpha = (float)((double)(urf_acosf(bracket) * 180) / M_PI);   /* :63 */

                if (alpha <= prm->angleFilter2 &&
                    (max1 - fabsf(a[j].z) >= prm->curbHeight ||
                     max2 - fabsf(a[j].z) >= prm->curbHeight) &&
                    (double)fabsf(max1 - max2) >= 0.05) {           /* :66-69 */
                    a[j].isCurbPoint = 2;                           /* :71 */
                    a[j].detect |= 4;
                }
            }
        }
    }
}

/* lidar_segmentation.cpp:70-93 partition / quickSort (Lomuto, pivot = last),
 * with an explicit stack instead of recursion (the sub-ranges are disjoint,
 * so the order in which they are processed does not change the result). */
static void quick_sort(pt3* a, int n)
{
    if (n < 2)
        return;
    int cap = 64, top = 0;
    int* st = (int*)malloc((size_t)cap * 2 * sizeof(int));
    st[0] = 0; st[1] = n - 1; top = 1;
    while (top > 0) {
        --top;
        int low = st[2 * top], high = st[2 * top + 1];
        if (low < high) {
            float pivot = a[high].alpha;
            int i = low - 1;
            for (int j = low; j <= high - 1; j++) {
                if (a[j].alpha < pivot) {
                    i++;
                    pt3 t = a[i]; a[i] = a[j]; a[j] = t;
                }
            }
            { pt3 t = a[i + 1]; a[i + 1] = a[high]; a[high] = t; }
            int pi = i + 1;
            if (top + 2 > cap) {
                cap *= 2;
                st = (int*)realloc(st, (size_t)cap * 2 * sizeof(int));
            }
            st[2 * top] = low;      st[2 * top + 1] = pi - 1; top++;
            st[2 * top] = pi + 1;   st[2 * top + 1] = high;   top++;
        }
    }
    free(st);
}

/* blind_spots.cpp:7-284 */
static void blind_spots(pt3** array3D, int index, const int* indexArray, const float* maxDistance,
                        const urf_params* prm, int channels, float* dbg_q, int16_t* dbg_stop)
{
    float q1 = 0, q2 = 180, q3 = 180, q4 = 360;
    int i, j, k, l;
    const float beamZone = prm->beamZone;

    if (dbg_stop)
        for (i = 0; i < 2 * 361; i++)
            dbg_stop[i] = -1;

    if (prm->blind_spots && channels > 1) {   /* :17-57 (deviation D4) */
        const pt3* a1 = array3D[1];
        for (i = 0; i < indexArray[1]; i++) {
            if (a1[i].isCurbPoint == 2) {
                if (a1[i].alpha >= 0 && a1[i].alpha < 90) {
                    if (a1[i].alpha > q1) q1 = a1[i].alpha;
                } else if (a1[i].alpha >= 90 && a1[i].alpha < 180) {
                    if (a1[i].alpha < q2) q2 = a1[i].alpha;
                } else if (a1[i].alpha >= 180 && a1[i].alpha < 270) {
                    if (a1[i].alpha > q3) q3 = a1[i].alpha;
                } else {
                    if (a1[i].alpha < q4) q4 = a1[i].alpha;
                }
            }
        }
    }
    if (dbg_q) { dbg_q[0] = q1; dbg_q[1] = q2; dbg_q[2] = q3; dbg_q[3] = q4; }

    float arcDistance;
    int notRoad;
    int blindSpot;
    float currentDegree;

    arcDistance = (float)((((double)maxDistance[0] * M_PI) / 180) * (double)beamZone);   /* :65 */

    for (i = 0; (float)i <= 360 - beamZone; i++) {   /* :68 */
        blindSpot = 0;
        if (prm->blind_spots) {                      /* :72-99 */
            if (prm->xDirection == 0) {
                if ((q1 != 0 && q4 != 360 && ((float)i <= q1 || (float)i >= q4)) ||
                    (q2 != 180 && q3 != 180 && (float)i >= q2 && (float)i <= q3))
                    blindSpot = 1;
            } else if (prm->xDirection == 1) {
                if ((q2 != 180 && (float)i >= q2 && i <= 270) || (q1 != 0 && ((float)i <= q1 || i >= 270)))
                    blindSpot = 1;
            } else {
                if ((q4 != 360 && ((float)i >= q4 || i <= 90)) || (q3 != 180 && (float)i <= q3 && i >= 90))
                    blindSpot = 1;
            }
        }
        if (blindSpot == 0) {
            notRoad = 0;
            const int n0 = indexArray[0];
            pt3* a0 = array3D[0];
            for (j = 0; j < n0 && a0[j].alpha <= (float)i + beamZone; j++) {   /* :107 (D3) */
                if (a0[j].alpha >= (float)i) {
                    if (a0[j].isCurbPoint == 2) {
                        notRoad = 1;
                        break;
                    }
                }
            }
            if (dbg_stop)
                dbg_stop[i] = (int16_t)(notRoad ? 0 : index);
            if (notRoad == 0) {
                for (j = 0; j < n0 && a0[j].alpha <= (float)i + beamZone; j++) {   /* :124 */
                    if (a0[j].alpha >= (float)i)
                        a0[j].isCurbPoint = 1;
                }
                for (k = 1; k < index; k++) {   /* :133 */
                    if ((float)i == 360 - beamZone)
                        currentDegree = 360;
                    else
                        currentDegree = (float)((double)i + (double)arcDistance / (((double)maxDistance[k] * M_PI) / 180));   /* :142 */
                    const int nk = indexArray[k];
                    pt3* ak = array3D[k];
                    for (l = 0; l < nk && ak[l].alpha <= currentDegree; l++) {   /* :146 */
                        if (ak[l].alpha >= (float)i) {
                            if (ak[l].isCurbPoint == 2) {
                                notRoad = 1;
                                break;
                            }
                        }
                    }
                    if (notRoad == 1) {
                        if (dbg_stop)
                            dbg_stop[i] = (int16_t)k;
                        break;
                    }
                    for (l = 0; l < nk && ak[l].alpha <= currentDegree; l++) {   /* :164 */
                        if (ak[l].alpha >= (float)i)
                            ak[l].isCurbPoint = 1;
                    }
                }
            }
        }
    }

    for (i = 360; (float)i >= 0 + beamZone; --i) {   /* :177 */
        blindSpot = 0;
        if (prm->blind_spots) {                      /* :181-208 */
            if (prm->xDirection == 0) {
                if ((q1 != 0 && q4 != 360 && ((float)i <= q1 || (float)i >= q4)) ||
                    (q2 != 180 && q3 != 180 && (float)i >= q2 && (float)i <= q3))
                    blindSpot = 1;
            } else if (prm->xDirection == 1) {
                if ((q2 != 180 && (float)i >= q2 && i <= 270) || (q1 != 0 && ((float)i <= q1 || i >= 270)))
                    blindSpot = 1;
            } else {
                if ((q4 != 360 && ((float)i >= q4 || i <= 90)) || (q3 != 180 && (float)i <= q3 && i >= 90))
                    blindSpot = 1;
            }
        }
        if (blindSpot == 0) {
            notRoad = 0;
            const int n0 = indexArray[0];
            pt3* a0 = array3D[0];
            for (j = n0 - 1; j >= 0 && a0[j].alpha >= (float)i - beamZone; --j) {   /* :216 (D3) */
                if (a0[j].alpha <= (float)i) {
                    if (a0[j].isCurbPoint == 2) {
                        notRoad = 1;
                        break;
                    }
                }
            }
            if (dbg_stop)
                dbg_stop[361 + i] = (int16_t)(notRoad ? 0 : index);
            if (notRoad == 0) {
                for (j = n0 - 1; j >= 0 && a0[j].alpha >= (float)i - beamZone; --j) {   /* :233 */
                    if (a0[j].alpha <= (float)i)
                        a0[j].isCurbPoint = 1;
                }
                for (k = 1; k < index; k++) {   /* :242 */
                    if ((float)i == 0 + beamZone)
                        currentDegree = 0;
                    else
                        currentDegree = (float)((double)i - (double)arcDistance / (((double)maxDistance[k] * M_PI) / 180));   /* :251 */
                    const int nk = indexArray[k];
                    pt3* ak = array3D[k];
                    for (l = nk - 1; l >= 0 && ak[l].alpha >= currentDegree; --l) {   /* :255 */
                        if (ak[l].alpha <= (float)i) {
                            if (ak[l].isCurbPoint == 2) {
                                notRoad = 1;
                                break;
                            }
                        }
                    }
                    if (notRoad == 1) {
                        if (dbg_stop)
                            dbg_stop[361 + i] = (int16_t)k;
                        break;
                    }
                    for (l = nk - 1; l >= 0 && ak[l].alpha >= currentDegree; --l) {   /* :273 */
                        if (ak[l].alpha <= (float)i)
                            ak[l].isCurbPoint = 1;
                    }
                }
            }
        }
    }
}

static int float_cmp(const void* a, const void* b)
{
    float fa = *(const float*)a, fb = *(const float*)b;
    return (fa > fb) - (fa < fb);
}

/* lidar_segmentation.cpp:95-293, 353-367, 605-608 Detector::filtered */
int urf_oracle_classify(const float* x, const float* y, const float* z, uint32_t n,
                        const urf_params* prm, uint8_t* labels,
                        urf_scan_info* info, urf_oracle_debug* dbg)
{
    if (!x || !y || !z || !prm || !labels)
        return URF_ERR_INVALID_ARG;
    if (prm->size != sizeof(urf_params) || prm->channels < 1 || prm->channels > 1024 ||
        prm->curbPoints < 1 || prm->sectors < 1)
        return URF_ERR_PARAMS;
    const int channels = prm->channels;
    uint32_t i;
    int j;

    memset(labels, 0, n);
    if (info)
        memset(info, 0, sizeof(*info));
    if (dbg) {
        for (i = 0; i < n; i++) {
            if (dbg->valpha) dbg->valpha[i] = -1.0f;
            if (dbg->ring) dbg->ring[i] = -1;
            if (dbg->azimuth) dbg->azimuth[i] = 0;
            if (dbg->range2d) dbg->range2d[i] = 0;
            if (dbg->detect) dbg->detect[i] = 0;
            if (dbg->sector) dbg->sector[i] = -1;
        }
        if (dbg->angle_table) memset(dbg->angle_table, 0, sizeof(float) * (size_t)channels);
        if (dbg->max_dist) memset(dbg->max_dist, 0, sizeof(float) * (size_t)channels);
        if (dbg->quadrants) { dbg->quadrants[0] = 0; dbg->quadrants[1] = 180; dbg->quadrants[2] = 180; dbg->quadrants[3] = 360; }
        if (dbg->beam_stop) for (j = 0; j < 2 * 361; j++) dbg->beam_stop[j] = -1;
    }

    /* :100-117 ROI filter (order preserving); NaN coordinates fail the compares */
    pt3* array2D = (pt3*)calloc(n ? n : 1, sizeof(pt3));
    size_t piece = 0;
    for (i = 0; i < n; i++) {
        float px = x[i], py = y[i], pz = z[i];
        if (px >= prm->min_X && px <= prm->max_X &&
            py >= prm->min_Y && py <= prm->max_Y &&
            pz >= prm->min_Z && pz <= prm->max_Z &&
            px + py + pz != 0) {
            array2D[piece].x = px; array2D[piece].y = py; array2D[piece].z = pz;
            array2D[piece].src = i;
            piece++;
        }
    }
    if (info)
        info->n_roi = (uint32_t)piece;

    if (piece < 30) {   /* :124-126 */
        free(array2D);
        if (info)
            info->status = URF_TOO_FEW_POINTS;
        return URF_TOO_FEW_POINTS;
    }

    float bracket;
    float* angle = (float*)calloc((size_t)channels, sizeof(float));   /* :136 */
    int index = 0;
    int newCircle;

    for (i = 0; i < piece; i++) {   /* :145-197 */
        pt3* q = &array2D[i];
        q->d = (float)sqrt((double)q->x * (double)q->x + (double)q->y * (double)q->y + (double)q->z * (double)q->z);   /* :148 */
        bracket = fabsf(q->z) / q->d;   /* :151 */
        if (bracket < -1)
            bracket = -1;
        else if (bracket > 1)
            bracket = 1;
        if (q->z < 0)
            q->alpha = (float)((double)(urf_acosf(bracket) * 180) / M_PI);          /* :162 */
        else
            q->alpha = (float)(((double)(urf_asinf(bracket) * 180) / M_PI) + 90);   /* :165 */
        if (dbg && dbg->valpha)
            dbg->valpha[q->src] = q->alpha;

        newCircle = 1;
        for (j = 0; j < channels; j++) {   /* :174-184 */
            if (angle[j] == 0)
                break;
            if (fabsf(angle[j] - q->alpha) <= prm->interval) {
                newCircle = 0;
                break;
            }
        }
        if (newCircle == 1) {   /* :187-196 */
            if (index < channels) {
                angle[index] = q->alpha;
                index++;
            }
        }
    }

    if (prm->star_shaped_method)   /* :199-200 */
        star_shaped_search(array2D, (int)piece, prm, dbg ? dbg->sector : NULL);

    qsort(angle, (size_t)index, sizeof(float), float_cmp);   /* :205 */
    if (dbg && dbg->angle_table)
        memcpy(dbg->angle_table, angle, sizeof(float) * (size_t)index);

    /* :207 array3D(channels, vector<Point3D>(piece)): here ring buckets sized on demand */
    int* indexArray = (int*)calloc((size_t)channels, sizeof(int));       /* :212 */
    float* maxDistance = (float*)calloc((size_t)channels, sizeof(float)); /* :215 */
    int16_t* ring_of = (int16_t*)malloc(piece * sizeof(int16_t));
    for (i = 0; i < piece; i++) {
        int results = 0;
        for (j = 0; j < index; j++) {   /* :226-233 */
            if (fabsf(angle[j] - array2D[i].alpha) <= prm->interval) {
                results = 1;
                break;
            }
        }
        ring_of[i] = (int16_t)(results ? j : -1);
        if (results)
            indexArray[j]++;
    }
    pt3** array3D = (pt3**)calloc((size_t)channels, sizeof(pt3*));
    for (j = 0; j < channels; j++) {
        array3D[j] = (pt3*)calloc((size_t)(indexArray[j] ? indexArray[j] : 1), sizeof(pt3));
        indexArray[j] = 0;
    }
    for (i = 0; i < piece; i++) {   /* :221-278 */
        if (ring_of[i] < 0)
            continue;
        j = ring_of[i];
        pt3* q = &array3D[j][indexArray[j]];
        q->x = array2D[i].x; q->y = array2D[i].y; q->z = array2D[i].z;   /* :238 */
        q->src = array2D[i].src;
        if (prm->star_shaped_method) {   /* :241-242 */
            q->isCurbPoint = array2D[i].isCurbPoint;
            q->detect = array2D[i].detect;
        }
        q->d = (float)sqrt((double)q->x * (double)q->x + (double)q->y * (double)q->y);   /* :245 */
        bracket = fabsf(q->x) / q->d;   /* :248 */
        if (bracket < -1)
            bracket = -1;
        else if (bracket > 1)
            bracket = 1;
        {
            double t = (double)(urf_asinf(bracket) * 180) / M_PI;
            if (q->x >= 0 && q->y <= 0)
                q->alpha = (float)t;           /* :256 */
            else if (q->x >= 0 && q->y > 0)
                q->alpha = (float)(180 - t);   /* :260 */
            else if (q->x < 0 && q->y >= 0)
                q->alpha = (float)(180 + t);   /* :264 */
            else
                q->alpha = (float)(360 - t);   /* :268 */
        }
        if (q->d > maxDistance[j])   /* :271-274 */
            maxDistance[j] = q->d;
        if (dbg) {
            if (dbg->ring) dbg->ring[q->src] = (int16_t)j;
            if (dbg->azimuth) dbg->azimuth[q->src] = q->alpha;
            if (dbg->range2d) dbg->range2d[q->src] = q->d;
        }
        indexArray[j]++;   /* :276 */
    }
    if (dbg && dbg->max_dist)
        memcpy(dbg->max_dist, maxDistance, sizeof(float) * (size_t)channels);

    if (prm->x_zero_method)   /* :280-281 */
        x_zero_method(array3D, index, indexArray, prm);
    if (prm->z_zero_method)   /* :282-283 */
        z_zero_method(array3D, index, indexArray, prm);

    if (dbg && dbg->detect) {
        for (j = 0; j < index; j++)
            for (int k = 0; k < indexArray[j]; k++)
                dbg->detect[array3D[j][k].src] = array3D[j][k].detect;
    }

    for (j = 0; j < index; j++)   /* :289-291 */
        quick_sort(array3D[j], indexArray[j]);

    blind_spots(array3D, index, indexArray, maxDistance, prm, channels,
                dbg ? dbg->quadrants : NULL, dbg ? dbg->beam_stop : NULL);   /* :293 */

    /* :295-351 marker points: per integer degree the farthest road point met before the first
     * non-road point of that degree, rings in order, every ring in azimuth order */
    if (dbg && dbg->marker_pts && dbg->n_marker_pts) {
        uint32_t cM = 0;
        for (int deg = 0; deg <= 360; deg++) {
            int ID1 = -1, ID2 = -1, redPoints = 0;
            float maxDistanceRoad = 0, d;
            for (j = 0; j < index; j++) {
                for (int k = 0; k < indexArray[j]; k++) {
                    const pt3* q = &array3D[j][k];
                    if (q->isCurbPoint != 1 && q->alpha >= (float)deg && q->alpha < (float)(deg + 1)) {   /* :318 */
                        redPoints = 1;
                        break;
                    }
                    if (q->isCurbPoint == 1 && q->alpha >= (float)deg && q->alpha < (float)(deg + 1)) {   /* :325 */
                        d = (float)sqrt((double)(0 - q->x) * (double)(0 - q->x) + (double)(0 - q->y) * (double)(0 - q->y));
                        if (d > maxDistanceRoad) {
                            maxDistanceRoad = d;
                            ID1 = j;
                            ID2 = k;
                        }
                    }
                }
                if (redPoints == 1)
                    break;
            }
            if (ID1 != -1 && ID2 != -1) {   /* :343-350 */
                dbg->marker_pts[4 * cM + 0] = array3D[ID1][ID2].x;
                dbg->marker_pts[4 * cM + 1] = array3D[ID1][ID2].y;
                dbg->marker_pts[4 * cM + 2] = array3D[ID1][ID2].z;
                dbg->marker_pts[4 * cM + 3] = (float)redPoints;
                cM++;
            }
        }
        *dbg->n_marker_pts = cM;
    }

    /* :354-367 road / curb; :620 roi; :605-608 road_probably */
    uint32_t n_road = 0, n_curb = 0, n_ring = 0, n_ring10 = 0;
    for (i = 0; i < piece; i++)
        labels[array2D[i].src] = URF_FLAG_ROI;
    for (j = 0; j < index; j++) {
        for (int k = 0; k < indexArray[j]; k++) {
            const pt3* q = &array3D[j][k];
            uint8_t l = URF_FLAG_ROI | URF_FLAG_RING;
            if (q->isCurbPoint == 1) { l |= URF_LABEL_ROAD; n_road++; }
            else if (q->isCurbPoint == 2) { l |= URF_LABEL_CURB; n_curb++; }
            if (q->isCurbPoint == 1 && dbg && dbg->road_order) dbg->road_order[n_road - 1] = q->src;
            if (q->isCurbPoint == 2 && dbg && dbg->curb_order) dbg->curb_order[n_curb - 1] = q->src;
            if (j == 10 && channels > 10) {   /* deviation D4 */
                if (dbg && dbg->ring10_order) dbg->ring10_order[n_ring10] = q->src;
                l |= URF_FLAG_RING10;
                n_ring10++;
            }
            labels[q->src] = l;
            n_ring++;
        }
    }
    if (info) {
        info->status = URF_OK;
        info->n_rings = (uint32_t)index;
        info->n_ring_pts = n_ring;
        info->n_road = n_road;
        info->n_curb = n_curb;
        info->n_ring10 = n_ring10;
    }

    for (j = 0; j < channels; j++)
        free(array3D[j]);
    free(array3D);
    free(ring_of);
    free(maxDistance);
    free(indexArray);
    free(angle);
    free(array2D);
    return URF_OK;
}

/* ---- lidar_segmentation.cpp:369-602: line strips from the marker points ------------------ */
typedef struct {
    int32_t id, action;
    float r, g, b, a;
    double pts[3 * 1024];
    int n;
} strip;

static void strip_push(strip* s, double x, double y, double z)
{
    if (s->n < 1024) {
        s->pts[3 * s->n] = x; s->pts[3 * s->n + 1] = y; s->pts[3 * s->n + 2] = z;
        s->n++;
    }
}
static void line_push(urf_oracle_marker_state* st, float x, float y)
{
    if (st->line_n < 1024) {
        st->line_x[st->line_n] = x; st->line_y[st->line_n] = y;
        st->line_n++;
    }
}
static int emit(urf_oracle_markers* out, const strip* s)
{
    if (out->n_markers >= out->cap_markers || out->n_points + (uint32_t)s->n > out->cap_points)
        return -1;
    urf_oracle_marker* m = &out->markers[out->n_markers++];
    m->id = s->id; m->action = s->action; m->type = 4;
    m->r = s->r; m->g = s->g; m->b = s->b; m->a = s->a;
    m->first_point = out->n_points; m->n_points = (uint32_t)s->n;
    memcpy(out->pts + 3 * out->n_points, s->pts, sizeof(double) * 3 * (size_t)s->n);
    out->n_points += (uint32_t)s->n;
    return 0;
}
/* :471-485 etc.: replace the strip's points by the simplified member line, z = polyz */
static void simplify_into(strip* s, const urf_oracle_marker_state* st, const urf_marker_params* mp)
{
    unsigned char keep[1024];
    s->n = 0;
    urf_rdp_float(st->line_x, st->line_y, st->line_n, mp->poly_s_param, keep);
    for (int i = 0; i < st->line_n; i++)
        if (keep[i])
            strip_push(s, (double)st->line_x[i], (double)st->line_y[i], (double)mp->poly_z_manual);
}

int urf_oracle_marker_strips(const float* mpts, uint32_t cM_in, const urf_marker_params* mp,
                             urf_oracle_marker_state* st, urf_oracle_markers* out)
{
    if (!mpts || !mp || !st || !out || mp->size != sizeof(*mp) || cM_in > 361)
        return URF_ERR_INVALID_ARG;
    const int cM = (int)cM_in;
    out->n_markers = 0;
    out->n_points = 0;
    out->published = 0;
    if (cM <= 2)   /* :371 */
        return URF_OK;
    float flag[361];
    int i;
    for (i = 0; i < cM; i++)
        flag[i] = mpts[4 * i + 3];
    if (flag[0] == 0 && flag[1] == 1) flag[0] = 1;                     /* :381 */
    if (flag[cM - 1] == 0 && flag[cM - 2] == 1) flag[cM - 1] = 1;      /* :386 */
    if (flag[0] == 1 && flag[1] == 0) flag[0] = 0;                     /* :391 */
    if (flag[cM - 1] == 1 && flag[cM - 2] == 0) flag[cM - 1] = 0;      /* :396 */
    for (i = 2; i <= cM - 3; i++)                                      /* :402 */
        if (flag[i] == 0 && flag[i - 1] == 1 && flag[i + 1] == 1) flag[i] = 1;
    for (i = 2; i <= cM - 3; i++)                                      /* :411 */
        if (flag[i] == 1 && flag[i - 1] == 0 && flag[i + 1] == 0) flag[i] = 0;

    strip ls;
    memset(&ls, 0, sizeof(ls));
    ls.action = 0;   /* ADD, :427 */
    float zavg = 0.0f;
    int lineStripID = 0;
    int rc = 0;
    for (i = 0; i < cM; i++) {   /* :430 */
        const double px = (double)mpts[4 * i], py = (double)mpts[4 * i + 1], pz = (double)mpts[4 * i + 2];
        zavg *= (float)i;                         /* :436-438 */
        zavg = (float)((double)zavg + pz);
        zavg /= (float)(i + 1);
        if (i == 0) {                             /* :442 */
            strip_push(&ls, px, py, pz);
            line_push(st, (float)px, (float)py);
        } else if (flag[i] == flag[i - 1]) {      /* :450 */
            strip_push(&ls, px, py, pz);
            line_push(st, (float)px, (float)py);
            if (i == cM - 1) {
                ls.id = lineStripID;
                if (flag[i] == 0) { ls.r = 0; ls.g = 1; ls.b = 0; ls.a = 1; }
                else { ls.r = 1; ls.g = 0; ls.b = 0; ls.a = 1; }
                if (mp->simple_poly_allow)
                    simplify_into(&ls, st, mp);
                rc |= emit(out, &ls);
                ls.n = 0;
                st->line_n = 0;
            }
        } else if (flag[i] == 0) {                /* :495 red -> green */
            strip_push(&ls, px, py, pz);
            line_push(st, (float)px, (float)py);
            ls.id = lineStripID;
            lineStripID++;
            ls.r = 1; ls.g = 0; ls.b = 0; ls.a = 1;
            if (mp->simple_poly_allow)
                simplify_into(&ls, st, mp);
            rc |= emit(out, &ls);
            ls.n = 0;
            st->line_n = 0;
            strip_push(&ls, px, py, pz);
            line_push(st, (float)px, (float)py);
        } else {                                  /* :534 green -> red */
            ls.id = lineStripID;
            lineStripID++;
            ls.r = 0; ls.g = 1; ls.b = 0; ls.a = 1;
            if (mp->simple_poly_allow)
                simplify_into(&ls, st, mp);
            rc |= emit(out, &ls);
            ls.n = 0;
            st->line_n = 0;
            const double qx = (double)mpts[4 * (i - 1)], qy = (double)mpts[4 * (i - 1) + 1], qz = (double)mpts[4 * (i - 1) + 2];
            strip_push(&ls, qx, qy, qz);
            line_push(st, (float)qx, (float)qy);
            strip_push(&ls, px, py, pz);
            line_push(st, (float)px, (float)py);
        }
    }
    if (mp->poly_z_avg_allow)                     /* :580-589 */
        for (uint32_t k = 0; k < out->n_points; k++)
            out->pts[3 * k + 2] = (double)zavg;
    ls.action = 2;                                /* DELETE, :592-597 */
    for (int del = lineStripID; del < st->ghostcount; del++) {
        ls.id++;
        rc |= emit(out, &ls);
    }
    st->ghostcount = lineStripID;                 /* :598 */
    out->published = 1;                           /* :601 */
    return rc ? URF_ERR_CAPACITY : URF_OK;
}
